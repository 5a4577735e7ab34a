// rc_model.h -- LDS layout of the adaptive order-0 models (gatb RangeCoder.cpp Order0Model [RECALLED]) shared by the
// block range coder (rc_kernels.hip) and the block decoder (decode_kernels.hip).
#pragma once
#include "leon_device.h"

namespace leon {

constexpr uint32_t RC_NSLOT_BIG = 24;        // numeric models cached in LDS per block (5 blocks per CU) ...
constexpr uint32_t RC_NSLOT_SMALL = 13;      // ... or, when there are more blocks than that keeps resident, 8 per CU
constexpr uint32_t RC_LW = 20;               // word offset of Lw[] inside a model
constexpr uint32_t RC_STRIDE = 280;          // 256-ary model: H[17] pad Lw[256] + one zero word (F(256) = H[16] + 0)
constexpr uint32_t RC_SSTRIDE = 40;          // small model (alphabet <= 5): same layout, only the first 16-block
constexpr uint32_t RC_SMALL_WORDS = N_SMALL_MODELS * RC_SSTRIDE;
constexpr uint32_t RC_CMP_WORDS = N_NUM_GROUPS * RC_SSTRIDE;   // the encoder's byte-count models in the small layout (rc_kernels.hip RcModeler<.., CMP>)
constexpr uint32_t RC_NNUM = N_NUM_GROUPS * MODELS_PER_NUMERIC;   // 72
constexpr uint32_t RC_GLOBAL = 0x80000000u;  // model lives in the global overflow area (more than RC_NSLOT numeric models)
constexpr uint64_t RC_BOTTOM = 1ull << 48;
constexpr uint32_t RC_RING = 65;
constexpr uint32_t RC_MAX_TOTAL = 1u << 30;  // the coder's multiply-high + 32-bit fix-up needs totals below this; from there on it divides exactly (per tile)


// A model keeps the cumulative count F(x) = H[x>>4] + Lw[x], x in 0..256 (F(256) = H[16] + the zero word).
// Order0Model::clear: F(x) = x.
template <typename P> __device__ inline void model_init(P s, uint32_t lane, bool small) {
    if (small) { if (lane < RC_SSTRIDE) s[lane] = (lane >= RC_LW && lane <= RC_LW + 16) ? lane - RC_LW : 0; }   // (Lw[16] = F(16): a byte-count model's total - 240)
    else {
        for (uint32_t x = lane; x <= 256; x += 64) s[RC_LW + x] = x < 256 ? (x & 15u) : 0u;
        if (lane < 17) s[lane] = 16 * lane;
    }
}

}  // namespace leon
