// OMPL-1.4.2-shaped interfaces for builds where OMPL is not installed (it is not in this image;
// art_planner/README.md:53 pins 1.4.2).  With -DARTP_HAVE_OMPL the real headers are used instead and this file
// declares nothing.
//
// The stand-ins are STRICT: every pure virtual of the real ompl::base::StateSampler (sampleUniform,
// sampleUniformNear, sampleGaussian), ompl::base::MotionValidator (both checkMotion overloads) and
// ompl::base::StateValidityChecker (isValid) is pure here too, with the real signatures, so a mirror class that
// would be abstract against the real library is abstract against this header as well and the stand-in build fails
// the same way.  Non-pure virtuals the mirror may override (clearance, the isValid overloads with distance) and
// the members the reference's code touches (si_, space_, rng_, valid_ / invalid_ counters, SE3 bounds) are there
// with OMPL's names.  Written from the published OMPL 1.4.2 headers (ompl/base/StateSampler.h,
// MotionValidator.h, StateValidityChecker.h, spaces/SE3StateSpace.h, spaces/RealVectorBounds.h, util/
// RandomNumbers.h); nothing of OMPL is executed here.
#pragma once

#ifdef ARTP_HAVE_OMPL
#include <ompl/base/MotionValidator.h>
#include <ompl/base/SpaceInformation.h>
#include <ompl/base/StateSampler.h>
#include <ompl/base/StateValidityChecker.h>
#include <ompl/base/spaces/SE3StateSpace.h>
#include <ompl/util/RandomNumbers.h>
#else
#include "art_planner/ompl_standins.h"
#endif
