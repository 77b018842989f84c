// art_planner::PlannerStatus, same enumerators and order as the reference
// (art_planner/include/art_planner/planner_status.h:9-16) so PlannerRos' switch statements keep working.
#pragma once

namespace art_planner {

enum PlannerStatus { UNKNOWN = 0, INVALID_START, INVALID_GOAL, NO_MAP, NOT_SOLVED, SOLVED };

}  // namespace art_planner
