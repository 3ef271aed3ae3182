// moc_zonal_nb5.hip -- instantiates the cdfmoc zonal kernel for a basin capacity of 5.
#define CDFT_MOC_NB 5
#include "moc_zonal.hpp"

namespace cdft {
namespace moc {
int launch_zonal_nb5(const ZonalArgs &a, int nwaves, int nv, bool aligned, int mode,
                     hipStream_t st) {
    return launch_nb<5>(a, nwaves, nv, aligned, mode, st);
}
}  // namespace moc
}  // namespace cdft
