// csrc/ingest.h — point-layout ingest (ingest.hip): strided point records -> packed float4 on the device.
#pragma once
#include "common.h"

namespace mrgfe {

int launch_gather_points(mrgfe_ctx* ctx, const void* d_raw, float4* d_dst, size_t n, uint32_t width, uint32_t row_step, uint32_t point_step, uint32_t ox, uint32_t oy, uint32_t oz,
                         int32_t oi);
int upload_gathered(mrgfe_ctx* ctx, const void* raw, size_t raw_bytes, size_t n, uint32_t width, uint32_t row_step, uint32_t point_step, uint32_t ox, uint32_t oy, uint32_t oz,
                    int32_t oi, void* d_dst);

}  // namespace mrgfe
