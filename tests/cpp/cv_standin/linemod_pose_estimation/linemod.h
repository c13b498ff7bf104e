// Stand-in for the header ROS's message generator (gencpp) makes from the reference's msg/linemod.msg (`int32 id`,
// `geometry_msgs/Transform tranform`) -- the generated file is not part of the reference repository and this image has no ROS.  It has the
// shape gencpp gives every message: a class template `linemod_<ContainerAllocator>`, `typedef ... linemod;`, `linemodPtr` /
// `linemodConstPtr`, and the message_traits specialisations with the type's name as a string.
// Why it is here: the service node includes rgbdDetector.h FIRST (src/linemod_ensenso_detect_3_mult_detect_service.cpp:1) and this header
// AFTER it (:16), so with the facade's `#define linemod lmx_linemod` active the typedef below is renamed.  The test compiles exactly that order
// (tests/cpp/cv_facade_main.cpp) and uses the message the way a node publishes one: the rename is consistent within the translation unit,
// `linemod_`, `linemodPtr` and the string literals are other tokens and stay as they are.
#ifndef LINEMOD_POSE_ESTIMATION_MESSAGE_LINEMOD_H
#define LINEMOD_POSE_ESTIMATION_MESSAGE_LINEMOD_H

#include <memory>
#include <stdint.h>

namespace boost { using std::shared_ptr; }   // stand-in: gencpp uses boost::shared_ptr
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
template <class ContainerAllocator> struct Transform_ { Vector3 translation; Quaternion rotation; };
typedef Transform_<std::allocator<void> > Transform;
}  // namespace geometry_msgs

namespace linemod_pose_estimation {
template <class ContainerAllocator>
struct linemod_ {
  typedef linemod_<ContainerAllocator> Type;
  linemod_() : id(0), tranform() {}
  explicit linemod_(const ContainerAllocator&) : id(0), tranform() {}
  typedef int32_t _id_type;
  _id_type id;
  typedef ::geometry_msgs::Transform_<ContainerAllocator> _tranform_type;
  _tranform_type tranform;
  typedef boost::shared_ptr< ::linemod_pose_estimation::linemod_<ContainerAllocator> > Ptr;
  typedef boost::shared_ptr< ::linemod_pose_estimation::linemod_<ContainerAllocator> const> ConstPtr;
};
typedef ::linemod_pose_estimation::linemod_<std::allocator<void> > linemod;
typedef boost::shared_ptr< ::linemod_pose_estimation::linemod > linemodPtr;
typedef boost::shared_ptr< ::linemod_pose_estimation::linemod const> linemodConstPtr;
}  // namespace linemod_pose_estimation

namespace ros {
namespace message_traits {
template <class M> struct DataType;
template <class ContainerAllocator>
struct DataType< ::linemod_pose_estimation::linemod_<ContainerAllocator> > {
  static const char* value() { return "linemod_pose_estimation/linemod"; }
  static const char* value(const ::linemod_pose_estimation::linemod_<ContainerAllocator>&) { return value(); }
};
}  // namespace message_traits
}  // namespace ros
#endif
