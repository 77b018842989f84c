// TEST SCAFFOLD, not OMPL: presents the planning-layer stand-ins under the real include name so that the
// -DARTP_HAVE_OMPL branch of the host mirror goes through a compiler in this image (OMPL is not installed here).
#pragma once
#include "art_planner/ompl_standins_planning.h"
