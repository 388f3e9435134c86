// placeholder until the fused kernel lands
#include "rph_internal.h"
int rph_launch_pdq_fused512(rph_ctx *, const uint8_t *, uint32_t, size_t, size_t, uint8_t *, float *, float *, uint8_t *,
                            uint8_t *, hipStream_t)
{
    return RPH_ERR_UNSUPPORTED;
}
