// Slab-distributed 3D operator over RCCL (one process per GPU).  Filled in below.
#include "plan.hpp"
namespace lsfc { DistState::~DistState() {} }
using namespace lsfc;
extern "C" {
int lsfc_dist_unique_id(unsigned char id[LSFC_UNIQUE_ID_BYTES]) {
    return guarded([&] { (void)id; fail(LSFC_EINVAL, "distributed plan not built yet"); });
}
int lsfc_dist_plan_create_gv3d(lsfc_plan** out, int64_t, int64_t, int64_t, double, double, const double*, unsigned, int, int, int,
                               const unsigned char*) {
    return guarded([&] { if (out) *out = nullptr; fail(LSFC_EINVAL, "distributed plan not built yet"); });
}
}
