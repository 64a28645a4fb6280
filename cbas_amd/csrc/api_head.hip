// placeholder until the head kernels land (replaced in the next commit)
#include "api_common.h"
extern "C" int64_t cbas_head_weights_count(const cbas_head_config*) { return -1; }
extern "C" int cbas_head_create(const cbas_head_config*, const float*, int64_t, int, cbas_head**) { return cbas_fail(CBAS_EINVAL, "head not built"); }
extern "C" void cbas_head_destroy(cbas_head*) {}
extern "C" int cbas_head_forward_windows(cbas_head*, const float*, int64_t, float*, float*, void*) { return cbas_fail(CBAS_EINVAL, "head not built"); }
extern "C" int cbas_head_infer_f16(cbas_head*, const uint16_t*, int64_t, float, float*, float*, void*) { return cbas_fail(CBAS_EINVAL, "head not built"); }
