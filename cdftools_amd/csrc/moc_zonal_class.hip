// moc_zonal_class.hip -- instantiates the class-mode cdfmoc zonal kernel (one LDS accumulator per
// basin-bit pattern, up to kMaxClass patterns).
#define CDFT_MOC_CLASS 1
#include "moc_zonal.hpp"

namespace cdft {
namespace moc {
int launch_zonal_class(const ZonalArgs &a, int nwaves, int nv, hipStream_t st) {
    return launch_class<kMaxClass>(a, nwaves, nv, st);
}
}  // namespace moc
}  // namespace cdft
