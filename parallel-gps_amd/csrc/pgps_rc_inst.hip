// pgps_rc_inst.hip -- one explicit instantiation of the row-cooperative level-1 kernels per scalar type and state
// dimension (-DPGPS_RC_T=double|float -DPGPS_RC_D=d, d = 2..16), so the fully unrolled units compile in parallel.
// The discretisation kernel computes in fp64 whatever the series' type: it lives in the double units only.
#include "pgps_rc.hip.h"
#ifndef PGPS_RC_NO_DISC
#include "pgps_rcgrad.hip.h"
#endif

#ifndef PGPS_RC_T
#define PGPS_RC_T double
#endif

namespace pgps {
namespace rc {
template int launch_rc_level1<PGPS_RC_T, PGPS_RC_D>(pgps_ctx*, const RcArgsT<PGPS_RC_T>&, int);
template int launch_rc_ks<PGPS_RC_T, PGPS_RC_D>(pgps_ctx*, int, long, long, const PGPS_RC_T*, PGPS_RC_T*, int, long, const PGPS_RC_T*);
template int launch_rc_scan_blocked<PGPS_RC_T, PGPS_RC_D>(pgps_ctx*, int, long, PGPS_RC_T*, PGPS_RC_T*);
template int launch_rc_seg_carry<PGPS_RC_T, PGPS_RC_D>(pgps_ctx*, int, const PGPS_RC_T*, int, int, int, PGPS_RC_T*);
#ifndef PGPS_RC_NO_DISC
template int launch_rc_disc<PGPS_RC_D>(pgps_ctx*, long, const double*, const double*, const double*, double, double*, double*, int, long);
template int launch_rc_grad<PGPS_RC_D>(pgps_ctx*, const GradLtiArgs&, int);
#endif
}  // namespace rc
}  // namespace pgps
