// lsa_hostmath.h -- host-side pose algebra of the product: the loops' 6-dof control flow is a handful of 4x4 products
// per ICP iteration.  The arithmetic itself is lsa_posemath.h, one source for this side and for the device (which works
// out the same poses behind a solve when the next iteration is enqueued ahead of it).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include "../lsa_posemath.h"

namespace lsa
{
namespace host
{
using namespace posemath;
}  // namespace host
}  // namespace lsa
