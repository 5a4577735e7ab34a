// rc_kernels.hip -- block-parallel order-0 adaptive range coder (gatb RangeCoder.cpp: Order0Model,
// RangeEncoder::encode / flush [RECALLED]).  The models reset per read block, so blocks are the only
// independent streams the format has: one 2-wave workgroup per block, software-pipelined over tiles of 64 symbols.
//   wave 1 (modeler, lane = symbol): every symbol's (cumLow, freq, total) from the model state at the tile start
//     plus the earlier symbols of the tile, then Order0Model::update for the tile, then the reciprocal of total;
//   wave 0 (coder, wave-uniform on the scalar unit): the serial chain
//     range /= total; low += cumLow*range; range *= freq; renormalise
//     with one multiply-high by the precomputed reciprocal instead of a 64-bit division.
// Counts are exactly Order0Model's cumulative counts (its rescale cannot trigger: MAX_RANGE = 2^48 total).
#include "kernels.h"
#include <cstdlib>

namespace leon {

constexpr uint32_t RC_NSLOT_BIG = 24;        // numeric models cached in LDS per block (5 blocks per CU) ...
constexpr uint32_t RC_NSLOT_SMALL = 14;      // ... or, when there are more blocks than that keeps resident, 8 per CU
constexpr uint32_t RC_LW = 20;               // word offset of Lw[] inside a model
constexpr uint32_t RC_STRIDE = 280;          // 256-ary model: H[17] pad Lw[256] + one zero word (F(256) = H[16] + 0)
constexpr uint32_t RC_SSTRIDE = 40;          // small model (alphabet <= 5): same layout, only the first 16-block
constexpr uint32_t RC_SMALL_WORDS = N_SMALL_MODELS * RC_SSTRIDE;
constexpr uint32_t RC_NNUM = N_NUM_GROUPS * MODELS_PER_NUMERIC;   // 72
constexpr uint32_t RC_GLOBAL = 0x80000000u;  // model lives in the global overflow area (more than RC_NSLOT numeric models)
constexpr uint64_t RC_BOTTOM = 1ull << 48;
constexpr uint32_t RC_MAX_TOTAL = 1u << 30;  // chain arithmetic assumes total < 2^30 (checked by the host per block)

size_t rc_model_scratch_bytes(uint64_t n_blocks) {
    return (size_t)n_blocks * (RC_NNUM - RC_NSLOT_SMALL) * RC_STRIDE * sizeof(uint32_t);
}

// keep a wave-uniform 64-bit value in vector registers (so that arithmetic on it issues on the vector unit)
__device__ inline void pin_v(uint64_t& x) {
    uint32_t a = (uint32_t)x, b = (uint32_t)(x >> 32);
    asm volatile("" : "+v"(a), "+v"(b));
    x = ((uint64_t)b << 32) | a;
}
// A model keeps the cumulative count F(x) = H[x>>4] + Lw[x], x in 0..256 (F(256) = H[16] + the zero word).
// Order0Model::clear: F(x) = x.
template <typename P> __device__ inline void model_init(P s, uint32_t lane, bool small) {
    if (small) { if (lane < RC_SSTRIDE) s[lane] = (lane >= RC_LW && lane < RC_LW + 16) ? lane - RC_LW : 0; }
    else {
        for (uint32_t x = lane; x <= 256; x += 64) s[RC_LW + x] = x < 256 ? (x & 15u) : 0u;
        if (lane < 17) s[lane] = 16 * lane;
    }
}


template <uint32_t RC_NSLOT, bool VCHAIN>
__global__ void __launch_bounds__(128) k_rc_encode(const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks,
                                                  uint8_t* out, const uint64_t* out_off, uint64_t* out_size,
                                                  uint32_t* scratch, int* err) {
    __shared__ uint32_t models[RC_SMALL_WORDS + RC_NSLOT * RC_STRIDE];
    __shared__ uint32_t ring[2][5][64];
    __shared__ uint8_t slotmap[RC_NNUM];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform: keeps the coder chain on the scalar unit
    // per-lane constants of the branch-free Order0Model::update: lanes 0..15 own Lw of the symbol's 16-block,
    // lanes 16..32 own H[0..16]
    const bool is_lw = lane < 16;
    const uint32_t upd_lane = is_lw ? lane : (lane < 33 ? lane - 16 : 0);
    const uint32_t upd_base = is_lw ? RC_LW + lane : upd_lane;
    const uint32_t upd_blkmask = is_lw ? ~0u : 0u;

    for (uint64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const uint64_t s0 = blk_begin[b], s1 = blk_begin[b + 1];
        if (s1 - s0 >= RC_MAX_TOTAL - 256) {                    // totals must stay below 2^30 for the chain's 32-bit fix-up
            if (threadIdx.x == 0) { atomicExch(err, 2); out_size[b] = 0; }
            continue;
        }
        const uint64_t ntiles = (s1 - s0 + 63) / 64;
        const uint16_t* sym16 = (const uint16_t*)syms;
        uint32_t* gmodels = scratch + b * (uint64_t)(RC_NNUM - RC_NSLOT_SMALL) * RC_STRIDE;
        // coder state (wave 0)
        uint64_t low = 0, range = ~0ull, nout = 0, acc = 0;
        uint8_t* dst = out + out_off[b];
        const uint64_t cap = out_off[b + 1] - out_off[b];
        bool overflow = false;
        // modeler state (wave 1); the next tile's symbols are fetched one tile ahead (a global load costs ~2000 cycles)
        uint32_t nused = 0;
        uint32_t raw_next = (wave == 1 && lane < (uint32_t)((s1 - s0) < 64 ? (s1 - s0) : 64)) ? sym16[s0 + lane] : 0xFFFFu;
        if (wave == 1) {                                       // AbstractDnaCoder::startBlock
            for (uint32_t m = 0; m < N_SMALL_MODELS; m++) model_init(&models[m * RC_SSTRIDE], lane, true);
            for (uint32_t i = lane; i < RC_NNUM; i += 64) slotmap[i] = 255;
        }
        __syncthreads();

        for (uint64_t t = 0; t <= ntiles; t++) {
            if (wave == 1) {
                if (t < ntiles) {
                    // =================== modeler: tile t -> ring[t & 1] ===================
                    const uint64_t base = s0 + t * 64;
                    const uint32_t cnt = (uint32_t)((s1 - base) < 64 ? (s1 - base) : 64);
                    const bool act = lane < cnt;
                    const uint32_t raw = raw_next;
                    {
                        const uint64_t nb = base + 64;
                        raw_next = (nb < s1 && lane < (uint32_t)((s1 - nb) < 64 ? (s1 - nb) : 64)) ? sym16[nb + lane] : 0xFFFFu;
                    }
                    const uint32_t m = raw & 0xff, c = raw >> 8;
                    const uint32_t key = (m << 8) | c;
                    const bool numeric = act && m >= N_SMALL_MODELS;
                    // slots for numeric models first seen in this tile
                    uint32_t slot = numeric ? slotmap[m - N_SMALL_MODELS] : 0;
                    unsigned long long need = __ballot(numeric && slot == 255);
                    while (need) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(need);
                        const uint32_t mm = (uint32_t)__builtin_amdgcn_readlane((int)m, (int)l);
                        const uint32_t ns = nused++;
                        if (lane == 0) slotmap[mm - N_SMALL_MODELS] = (uint8_t)ns;
                        if (ns < RC_NSLOT) model_init(&models[RC_SMALL_WORDS + ns * RC_STRIDE], lane, false);
                        else { model_init(gmodels + (uint64_t)(ns - RC_NSLOT) * RC_STRIDE, lane, false); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                        if (numeric && m == mm) slot = ns;
                        need &= ~__ballot(numeric && m == mm);
                    }
                    // word offset of this lane's model (RC_GLOBAL | offset for the overflow area)
                    uint32_t mb = 0;
                    if (act) mb = !numeric ? m * RC_SSTRIDE
                                           : (slot < RC_NSLOT ? RC_SMALL_WORDS + slot * RC_STRIDE : (RC_GLOBAL | ((slot - RC_NSLOT) * RC_STRIDE)));
                    const uint32_t tot_idx = numeric ? 16u : RC_LW + small_model_size(m);
                    __builtin_amdgcn_wave_barrier();
                    // counts at the tile start
                    uint32_t lo = 0, hi = 1, tot = 1;
                    if (act) {
                        if (!(mb & RC_GLOBAL)) {
                            const uint32_t* s = &models[mb];
                            lo = s[c >> 4] + s[RC_LW + c]; hi = s[(c + 1) >> 4] + s[RC_LW + c + 1]; tot = s[tot_idx];
                        } else {
                            const uint32_t* s = gmodels + (mb & ~RC_GLOBAL);
                            lo = s[c >> 4] + s[RC_LW + c]; hi = s[(c + 1) >> 4] + s[RC_LW + c + 1]; tot = s[tot_idx];
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    // earlier symbols of the tile + Order0Model::update, symbol by symbol, branch-free
                    const uint32_t lowkey = m << 8;
                    for (uint32_t i = 0; i < cnt; i++) {
                        const uint32_t ki = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)i);
                        const uint32_t mbi = (uint32_t)__builtin_amdgcn_readlane((int)mb, (int)i);
                        const uint32_t ci = ki & 0xff;
                        uint32_t dd = ki - lowkey;                      // < 256 <=> same model; then dd = c_i
                        dd = lane > i ? dd : 0xFFFFFFFFu;
                        lo += dd < c ? 1u : 0u;
                        hi += dd <= c ? 1u : 0u;
                        tot += dd < 256u ? 1u : 0u;
                        const uint32_t thr = is_lw ? (ci & 15u) : (ci >> 4);
                        const uint32_t idx = upd_base + ((ci & ~15u) & upd_blkmask);
                        if (upd_lane > thr) {
                            if (!(mbi & RC_GLOBAL)) (void)__hip_atomic_fetch_add(&models[mbi + idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            else (void)__hip_atomic_fetch_add(gmodels + (mbi & ~RC_GLOBAL) + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                        if (mbi & RC_GLOBAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_wave_barrier();
                    const uint64_t inv = ~0ull / (uint64_t)tot;        // per lane, off the serial chain
                    uint32_t(*rg)[64] = ring[t & 1];
                    rg[0][lane] = lo; rg[1][lane] = hi - lo; rg[2][lane] = tot;
                    rg[3][lane] = (uint32_t)inv; rg[4][lane] = (uint32_t)(inv >> 32);
                }
            } else if (t > 0) {
                // =================== coder: tile t-1 from ring[(t-1) & 1] ===================
                const uint64_t base = s0 + (t - 1) * 64;
                const uint32_t cnt = (uint32_t)((s1 - base) < 64 ? (s1 - base) : 64);
                uint32_t(*rg)[64] = ring[(t - 1) & 1];
                const uint32_t v_lo = rg[0][lane], v_fr = rg[1][lane], v_tot = rg[2][lane], v_il = rg[3][lane], v_ih = rg[4][lane];
                for (uint32_t j = 0; j < cnt; j++) {
                    const uint32_t s_lo = (uint32_t)__builtin_amdgcn_readlane((int)v_lo, (int)j);
                    const uint32_t s_fr = (uint32_t)__builtin_amdgcn_readlane((int)v_fr, (int)j);
                    const uint32_t s_tot = (uint32_t)__builtin_amdgcn_readlane((int)v_tot, (int)j);
                    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)v_il, (int)j);
                    const uint32_t b1 = (uint32_t)__builtin_amdgcn_readlane((int)v_ih, (int)j);
                    if (VCHAIN) {
                        // The same chain on the VECTOR unit (every lane computes the same values): a CU has ONE scalar
                        // unit for all its waves, and with 8 blocks resident per CU eight scalar chains queue on it.
                        pin_v(range); pin_v(low);
                        const uint32_t vr0 = (uint32_t)range, vr1 = (uint32_t)(range >> 32);
                        uint64_t q = (uint64_t)vr1 * b1 + (((uint64_t)vr1 * b0) >> 32) + (((uint64_t)vr0 * b1) >> 32);
                        const uint32_t rem0 = vr0 - (uint32_t)q * s_tot;          // true remainder < 4 * tot < 2^32
                        const bool ge2 = rem0 >= 2 * s_tot;
                        const uint32_t rem1 = rem0 - (ge2 ? 2 * s_tot : 0u);
                        q += (ge2 ? 2u : 0u) + (rem1 >= s_tot ? 1u : 0u);
                        low += (uint64_t)s_lo * q;
                        range = q * s_fr;
                        for (;;) {                                      // RangeEncoder::encode's while loop, uniform branches
                            pin_v(range); pin_v(low);
                            const uint32_t xh = (uint32_t)((low ^ (low + range)) >> 32);
                            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)xh) >= (1u << 24)) {
                                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(range >> 32)) >= (1u << 16)) break;
                                range = (0 - low) & (RC_BOTTOM - 1);
                            }
                            acc = (acc << 8) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(low >> 56));
                            nout++;
                            if ((nout & 7) == 0) {
                                if (nout <= cap) { if (lane == 0) *(uint64_t*)(dst + nout - 8) = __builtin_bswap64(acc); }
                                else overflow = true;
                            }
                            range <<= 8;
                            low <<= 8;
                        }
                        continue;
                    }
                    // q = floor(range / tot): truncated multiply-high by the reciprocal (at most 3 short, since
                    // total < 2^30), fixed up on the low word; low += cumLow * q; range = q * freq.  Hand-scheduled on
                    // the scalar unit: 29 instructions (the compiler's version of the same C was ~45 and went
                    // through VALU compares).
                    uint32_t r0 = (uint32_t)range, r1 = (uint32_t)(range >> 32), l0 = (uint32_t)low, l1 = (uint32_t)(low >> 32);
                    uint32_t q0, q1, t0, t1, t2, t3;
                    asm volatile(
                        "s_mul_hi_u32 %[t0], %[r1], %[b0]\n\t"
                        "s_mul_hi_u32 %[t1], %[r0], %[b1]\n\t"
                        "s_mul_i32 %[q0], %[r1], %[b1]\n\t"
                        "s_mul_hi_u32 %[q1], %[r1], %[b1]\n\t"
                        "s_add_u32 %[q0], %[q0], %[t0]\n\t"
                        "s_addc_u32 %[q1], %[q1], 0\n\t"
                        "s_add_u32 %[q0], %[q0], %[t1]\n\t"
                        "s_addc_u32 %[q1], %[q1], 0\n\t"
                        "s_mul_i32 %[t0], %[q0], %[tot]\n\t"
                        "s_sub_u32 %[t0], %[r0], %[t0]\n\t"              // remainder, < 4 * tot
                        "s_lshl_b32 %[t1], %[tot], 1\n\t"
                        "s_cmp_ge_u32 %[t0], %[t1]\n\t"
                        "s_cselect_b32 %[t2], %[t1], 0\n\t"
                        "s_cselect_b32 %[t3], 2, 0\n\t"
                        "s_sub_u32 %[t0], %[t0], %[t2]\n\t"
                        "s_cmp_ge_u32 %[t0], %[tot]\n\t"
                        "s_addc_u32 %[t3], %[t3], 0\n\t"
                        "s_add_u32 %[q0], %[q0], %[t3]\n\t"
                        "s_addc_u32 %[q1], %[q1], 0\n\t"
                        "s_mul_i32 %[t0], %[lo], %[q0]\n\t"              // low += cumLow * q
                        "s_mul_hi_u32 %[t1], %[lo], %[q0]\n\t"
                        "s_mul_i32 %[t2], %[lo], %[q1]\n\t"
                        "s_add_u32 %[t1], %[t1], %[t2]\n\t"
                        "s_add_u32 %[l0], %[l0], %[t0]\n\t"
                        "s_addc_u32 %[l1], %[l1], %[t1]\n\t"
                        "s_mul_hi_u32 %[t0], %[q0], %[fr]\n\t"           // range = q * freq
                        "s_mul_i32 %[r1], %[q1], %[fr]\n\t"
                        "s_mul_i32 %[r0], %[q0], %[fr]\n\t"
                        "s_add_u32 %[r1], %[r1], %[t0]\n\t"
                        : [r0] "+s"(r0), [r1] "+s"(r1), [l0] "+s"(l0), [l1] "+s"(l1), [q0] "=&s"(q0), [q1] "=&s"(q1),
                          [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3)
                        : [b0] "s"(b0), [b1] "s"(b1), [tot] "s"(s_tot), [lo] "s"(s_lo), [fr] "s"(s_fr)
                        : "scc");
                    range = ((uint64_t)r1 << 32) | r0;
                    low = ((uint64_t)l1 << 32) | l0;
                    for (;;) {                                          // RangeEncoder::encode's while loop
                        uint32_t xh;                                    // high word of low ^ (low + range), on the scalar unit
                        {
                            uint32_t a0 = (uint32_t)low, a1 = (uint32_t)(low >> 32), c0 = (uint32_t)range, c1 = (uint32_t)(range >> 32), u0;
                            asm volatile("s_add_u32 %[u0], %[a0], %[c0]\n\t"
                                         "s_addc_u32 %[xh], %[a1], %[c1]\n\t"
                                         "s_xor_b32 %[xh], %[xh], %[a1]\n\t"
                                         : [u0] "=&s"(u0), [xh] "=&s"(xh) : [a0] "s"(a0), [a1] "s"(a1), [c0] "s"(c0), [c1] "s"(c1) : "scc");
                        }
                        if (xh >= (1u << 24)) {
                            if ((uint32_t)(range >> 32) >= (1u << 16)) break;
                            range = (0 - low) & (RC_BOTTOM - 1);
                        }
                        acc = (acc << 8) | (low >> 56);
                        nout++;
                        if ((nout & 7) == 0) {
                            if (nout <= cap) { if (lane == 0) *(uint64_t*)(dst + nout - 8) = __builtin_bswap64(acc); }
                            else overflow = true;
                        }
                        range <<= 8;
                        low <<= 8;
                    }
                }
            }
            __syncthreads();
        }
        if (wave == 0) {
            for (int i = 0; i < 8; i++) {                               // RangeEncoder::flush
                acc = (acc << 8) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(low >> 56));
                nout++;
                if ((nout & 7) == 0) {
                    if (nout <= cap) { if (lane == 0) *(uint64_t*)(dst + nout - 8) = __builtin_bswap64(acc); }
                    else overflow = true;
                }
                low <<= 8;
            }
            const uint32_t tail = (uint32_t)(nout & 7);                 // bytes still in acc
            if (tail) {
                if (nout <= cap) { if (lane < tail) dst[nout - tail + lane] = (uint8_t)(acc >> (8 * (tail - 1 - lane))); }
                else overflow = true;
            }
            if (lane == 0) {
                out_size[b] = nout;
                if (overflow) atomicExch(err, 1);
            }
        }
        __syncthreads();
    }
}

void launch_rc_encode(hipStream_t s, const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks, uint8_t* out,
                      const uint64_t* out_off, uint64_t* out_size, uint32_t* model_scratch, int* err) {
    if (!n_blocks) return;
    uint32_t g = n_blocks > 65535 ? 65535u : (uint32_t)n_blocks;
    const bool small = n_blocks > 256 * 5;                   // keep every block resident: 8 x 19.6 KB per CU
    // scalar chain while a CU holds a block or two, vector chain once several blocks would queue on its one scalar unit
    static const char* force = getenv("LEON_RC_CHAIN");       // "s" / "v": measurement override
    const bool vchain = force ? force[0] == 'v' : n_blocks > 512;
#define RC_LAUNCH(N, V) hipLaunchKernelGGL((k_rc_encode<N, V>), dim3(g), dim3(128), 0, s, syms, blk_begin, n_blocks, out, out_off, out_size, model_scratch, err)
    if (small) { if (vchain) RC_LAUNCH(RC_NSLOT_SMALL, true); else RC_LAUNCH(RC_NSLOT_SMALL, false); }
    else { if (vchain) RC_LAUNCH(RC_NSLOT_BIG, true); else RC_LAUNCH(RC_NSLOT_BIG, false); }
#undef RC_LAUNCH
}

}  // namespace leon
