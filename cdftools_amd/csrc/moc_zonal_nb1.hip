// moc_zonal_nb1.hip -- instantiates the cdfmoc zonal kernel for a basin capacity of 1.
#define CDFT_MOC_NB 1
#include "moc_zonal.hpp"

namespace cdft {
namespace moc {
int launch_zonal_nb1(const ZonalArgs &a, int nwaves, int nv, bool aligned, int mode,
                     hipStream_t st) {
    return launch_nb<1>(a, nwaves, nv, aligned, mode, st);
}
}  // namespace moc
}  // namespace cdft
