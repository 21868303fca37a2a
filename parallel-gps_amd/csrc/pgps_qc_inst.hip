// pgps_qc_inst.hip -- one explicit instantiation of the quad-cooperative level-1 kernels per state dimension
// (-DPGPS_QC_D=d, d = 5..8; fp32 only).
#include "pgps_qc.hip.h"

namespace pgps {
namespace qc {
template int launch_qc_level1<PGPS_QC_D>(pgps_ctx*, const rc::RcArgsT<float>&, int);
}  // namespace qc
}  // namespace pgps
