// Explicit instantiations of the large kernel families, one part per compilation (-DBTF_INST_PART=n; see
// btf_instances.h).  gfx950 only.
#ifndef BTF_INST_PART
#error "compile with -DBTF_INST_PART=<0..7>"
#endif
#include "btf_instances.h"
