// Stand-in for the header gencpp makes from the reference's srv/linemod_pose.srv (`int32 object_id` --- `geometry_msgs/Transform pose`),
// included by the service node right before linemod.h (src/linemod_ensenso_detect_3_mult_detect_service.cpp:15) and used as
// `linemod_pose_estimation::linemod_pose::Request` / `::Response` (:1779-1780).  `linemod_pose` is one token: the facade's macro does not touch it.
#ifndef LINEMOD_POSE_ESTIMATION_MESSAGE_LINEMOD_POSE_H
#define LINEMOD_POSE_ESTIMATION_MESSAGE_LINEMOD_POSE_H

#include "linemod_pose_estimation/linemod.h"

namespace linemod_pose_estimation {
template <class ContainerAllocator> struct linemod_poseRequest_ { int32_t object_id; linemod_poseRequest_() : object_id(0) {} };
template <class ContainerAllocator> struct linemod_poseResponse_ { ::geometry_msgs::Transform_<ContainerAllocator> pose; };
typedef linemod_poseRequest_<std::allocator<void> > linemod_poseRequest;
typedef linemod_poseResponse_<std::allocator<void> > linemod_poseResponse;
struct linemod_pose {
  typedef linemod_poseRequest Request;
  typedef linemod_poseResponse Response;
  Request request;
  Response response;
};
}  // namespace linemod_pose_estimation
#endif
