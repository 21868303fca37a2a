// pgps_rc_inst.hip -- one explicit instantiation of the row-cooperative level-1 kernels per state dimension
// (-DPGPS_RC_D=d, d = 2..16), so the fully unrolled units compile in parallel.
#include "pgps_rc.hip.h"

namespace pgps {
namespace rc {
template int launch_rc_level1<PGPS_RC_D>(pgps_ctx*, const RcArgs&, int);
template int launch_rc_ks<PGPS_RC_D>(pgps_ctx*, int, long, long, const double*, double*, int, long, const double*);
template int launch_rc_seg_carry<PGPS_RC_D>(pgps_ctx*, int, const double*, int, int, int, double*);
template int launch_rc_disc<PGPS_RC_D>(pgps_ctx*, long, const double*, const double*, const double*, double, double*, double*, int, long);
}  // namespace rc
}  // namespace pgps
