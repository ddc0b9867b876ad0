// placeholder until dp_kernels.hip lands
#include "lz_ctx.hpp"
extern "C" int lzgpu_gapped_extend(const lz_gapped_args*, lz_align** out, uint64_t* n_out, uint32_t** ops, uint64_t* n_ops)
{
    if (out) *out = nullptr; if (n_out) *n_out = 0; if (ops) *ops = nullptr; if (n_ops) *n_ops = 0;
    return LZGPU_NH_UNSUPPORTED;
}
