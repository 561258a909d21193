// LidarSlam/Enums.h -- slam_lib/include/LidarSlam/Enums.h (the values the hot path uses)
#pragma once
#include <map>
#include <string>
#include <vector>

namespace LidarSlam
{

enum Keypoint { EDGE = 0, PLANE = 1, BLOB = 2, nKeypointTypes };
static const std::vector<Keypoint> KeypointTypes = {EDGE, PLANE, BLOB};
static const std::map<Keypoint, std::string> KeypointTypeNames = {{EDGE, "edge"}, {PLANE, "plane"}, {BLOB, "blob"}};
enum UndistortionMode { NONE = 0, ONCE = 1, REFINED = 2 };
enum class EgoMotionMode { NONE = 0, MOTION_EXTRAPOLATION = 1, REGISTRATION = 2, MOTION_EXTRAPOLATION_AND_REGISTRATION = 3 };
enum class MappingMode { NONE = 0, ADD_KPTS_TO_FIXED_MAP = 1, UPDATE = 2 };
enum class SamplingMode { FIRST = 0, LAST = 1, MAX_INTENSITY = 2, CENTER_POINT = 3, CENTROID = 4 };

}  // namespace LidarSlam
