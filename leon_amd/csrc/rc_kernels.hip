// rc_kernels.hip -- block-parallel order-0 adaptive range coder (gatb RangeCoder.cpp: Order0Model,
// RangeEncoder::encode / flush [RECALLED]).  One wave per read block: the coder state (low, range) is
// wave-uniform, the lanes share the O(alphabet) cumulative-count update of Order0Model::update.
#include "kernels.h"

namespace leon {

constexpr uint32_t RC_NSLOT = 24;            // numeric models cached in LDS per wave (1040 B each)
constexpr uint32_t RC_STRIDE = 260;          // 257 cumulative counts, padded
constexpr uint32_t RC_NNUM = N_NUM_GROUPS * MODELS_PER_NUMERIC;   // 72
constexpr uint64_t RC_TOP = 1ull << 56, RC_BOTTOM = 1ull << 48;

size_t rc_model_scratch_bytes(uint64_t n_blocks) {
    return (size_t)n_blocks * (RC_NNUM - RC_NSLOT) * RC_STRIDE * sizeof(uint32_t);
}

struct RcState {
    uint64_t low, range, n;
    uint64_t cap;
    uint8_t* out;
    bool overflow;
};

__device__ inline void rc_apply(RcState& st, uint32_t lo, uint32_t hi, uint32_t tot, uint32_t lane) {
    st.range /= tot;
    st.low += (uint64_t)lo * st.range;
    st.range *= (uint64_t)(hi - lo);
    while ((st.low ^ (st.low + st.range)) < RC_TOP ||
           (st.range < RC_BOTTOM && ((st.range = (0 - st.low) & (RC_BOTTOM - 1)), true))) {
        if (st.n < st.cap) { if (lane == 0) st.out[st.n] = (uint8_t)(st.low >> 56); }
        else st.overflow = true;
        st.n++;
        st.range <<= 8;
        st.low <<= 8;
    }
}

__global__ void __launch_bounds__(64) k_rc_encode(const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks,
                                                 uint8_t* out, const uint64_t* out_off, uint64_t* out_size,
                                                 uint32_t* scratch, int* err) {
    __shared__ uint32_t small[N_SMALL_MODELS][8];
    __shared__ uint32_t slots[RC_NSLOT][RC_STRIDE];
    __shared__ uint8_t slotmap[RC_NNUM];
    uint32_t lane = threadIdx.x;
    for (uint64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        // AbstractDnaCoder::startBlock: every model back to _charRanges[i] = i
        if (lane < 8) for (uint32_t m = 0; m < N_SMALL_MODELS; m++) small[m][lane] = lane;
        for (uint32_t i = lane; i < RC_NNUM; i += 64) slotmap[i] = 255;
        uint32_t nused = 0;
        RcState st;
        st.low = 0; st.range = ~0ull; st.n = 0; st.overflow = false;
        st.out = out + out_off[b];
        st.cap = out_off[b + 1] - out_off[b];
        uint32_t* gmodels = scratch + b * (uint64_t)(RC_NNUM - RC_NSLOT) * RC_STRIDE;
        uint64_t s0 = blk_begin[b], s1 = blk_begin[b + 1];
        const uint16_t* sym16 = (const uint16_t*)syms;
        for (uint64_t base = s0; base < s1; base += 64) {
            uint32_t cnt = (uint32_t)((s1 - base) < 64 ? (s1 - base) : 64);
            uint32_t v = lane < cnt ? sym16[base + lane] : 0;
            for (uint32_t j = 0; j < cnt; j++) {
                uint32_t mv = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)j);
                uint32_t m = mv & 0xff, c = mv >> 8;
                uint32_t lo, hi, tot;
                if (m < N_SMALL_MODELS) {
                    uint32_t S = small_model_size(m);
                    uint32_t* arr = small[m];
                    lo = arr[c]; hi = arr[c + 1]; tot = arr[S];
                    if (lane <= S && lane > c) arr[lane] += 1;
                } else {
                    uint32_t mi = m - N_SMALL_MODELS;
                    uint32_t slot = slotmap[mi];
                    if (slot == 255) {                        // first use of this numeric model in the block
                        slot = nused++;
                        if (lane == 0) slotmap[mi] = (uint8_t)slot;
                        if (slot < RC_NSLOT) { for (uint32_t i = lane; i <= 256; i += 64) slots[slot][i] = i; }
                        else {
                            uint32_t* arr = gmodels + (uint64_t)(slot - RC_NSLOT) * RC_STRIDE;
                            for (uint32_t i = lane; i <= 256; i += 64) arr[i] = i;
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    }
                    if (slot < RC_NSLOT) {
                        uint32_t* arr = slots[slot];
                        lo = arr[c]; hi = arr[c + 1]; tot = arr[256];
                        for (uint32_t i = lane; i <= 256; i += 64) if (i > c) arr[i] += 1;
                    } else {
                        uint32_t* arr = gmodels + (uint64_t)(slot - RC_NSLOT) * RC_STRIDE;
                        lo = arr[c]; hi = arr[c + 1]; tot = arr[256];
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        for (uint32_t i = lane; i <= 256; i += 64) if (i > c) arr[i] += 1;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                rc_apply(st, lo, hi, tot, lane);
                __builtin_amdgcn_wave_barrier();
            }
        }
        // RangeEncoder::flush
        for (int i = 0; i < 8; i++) {
            if (st.n < st.cap) { if (lane == 0) st.out[st.n] = (uint8_t)(st.low >> 56); }
            else st.overflow = true;
            st.n++;
            st.low <<= 8;
        }
        if (lane == 0) {
            out_size[b] = st.n;
            if (st.overflow) atomicExch(err, 1);
        }
        __syncthreads();
    }
}

void launch_rc_encode(hipStream_t s, const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks, uint8_t* out,
                      const uint64_t* out_off, uint64_t* out_size, uint32_t* model_scratch, int* err) {
    if (!n_blocks) return;
    uint32_t g = n_blocks > 65535 ? 65535u : (uint32_t)n_blocks;
    hipLaunchKernelGGL(k_rc_encode, dim3(g), dim3(64), 0, s, syms, blk_begin, n_blocks, out, out_off, out_size, model_scratch, err);
}

}  // namespace leon
