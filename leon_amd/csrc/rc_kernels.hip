// rc_kernels.hip -- block-parallel order-0 adaptive range coder (gatb RangeCoder.cpp: Order0Model,
// RangeEncoder::encode / flush [RECALLED]).  The models reset per read block, so blocks are the only
// independent streams the format has.  A workgroup takes a group of up to 8 blocks, software-pipelined over tiles of 64 symbols:
//   one modeler wave per block (lane = symbol): every symbol's (cumLow, freq, total) from the model state at the tile
//     start plus the earlier symbols of the tile, then Order0Model::update for the tile, then the reciprocal of total;
//   one coder wave per group (lane = block): the serial chains
//     range /= total; low += cumLow*range; range *= freq; renormalise
//     with one multiply-high by the precomputed reciprocal instead of a 64-bit division.
// Counts are exactly Order0Model's cumulative counts (its rescale cannot trigger: MAX_RANGE = 2^48 total).
#include "kernels.h"
#include "rc_model.h"
#include <algorithm>
#include <cstdlib>

namespace leon {

size_t rc_model_scratch_bytes(uint64_t n_blocks) {
    return (size_t)n_blocks * (RC_NNUM - RC_NSLOT_SMALL) * RC_STRIDE * sizeof(uint32_t);
}

// Waves of a workgroup land on the CU's four SIMDs round-robin.  Wave 0 is the coder; the waves whose index is a multiple
// of 4 would share its SIMD and stay idle (they only keep the barriers), so the chain never waits for a modeler's
// vector instruction; the other waves are the modelers.
__host__ __device__ constexpr uint32_t rc_waves(uint32_t g) { return g + 1 + (g - 1) / 3; }

// The earlier symbols of a tile, lane = symbol: lo += #(same model, smaller symbol), hi += #(same model, symbol <= mine),
// tot += #(same model) over the lanes before this one -- 64 steps of a broadcast and three compare/add-with-carry pairs; the lane
// mask "after lane i" is exec itself, shifted once per step (one asm block, exec restored inside it).  key = model << 8 | symbol.
__device__ inline void rc_rank_loop(uint32_t key, uint32_t c, uint32_t& lo, uint32_t& hi, uint32_t& tot) {
    const uint32_t lowkey = key & ~0xFFu;
    uint32_t sk0, sk1, dd;
    uint64_t m1, m2, sav;
    asm volatile(
        "s_mov_b64 %[sav], exec\n\t"
        "v_readlane_b32 %[sk0], %[key], 0\n\t"
        "s_mov_b64 exec, -2\n\t"
        "s_nop 1\n\t"
        ".set rc_i, 0\n\t"
        ".rept 32\n\t"
        "v_sub_u32 %[dd], %[sk0], %[lowkey]\n\t"
        "v_readlane_b32 %[sk1], %[key], rc_i + 1\n\t"
        "v_cmp_lt_u32 vcc, %[dd], %[c]\n\t"
        "v_cmp_le_u32 %[m1], %[dd], %[c]\n\t"
        "v_cmp_gt_u32 %[m2], %[k256], %[dd]\n\t"
        "v_addc_co_u32 %[lo], vcc, 0, %[lo], vcc\n\t"
        "v_addc_co_u32 %[hi], %[m1], 0, %[hi], %[m1]\n\t"
        "v_addc_co_u32 %[tot], %[m2], 0, %[tot], %[m2]\n\t"
        "s_lshl_b64 exec, exec, 1\n\t"
        "v_sub_u32 %[dd], %[sk1], %[lowkey]\n\t"
        "v_readlane_b32 %[sk0], %[key], (rc_i + 2) & 63\n\t"
        "v_cmp_lt_u32 vcc, %[dd], %[c]\n\t"
        "v_cmp_le_u32 %[m1], %[dd], %[c]\n\t"
        "v_cmp_gt_u32 %[m2], %[k256], %[dd]\n\t"
        "v_addc_co_u32 %[lo], vcc, 0, %[lo], vcc\n\t"
        "v_addc_co_u32 %[hi], %[m1], 0, %[hi], %[m1]\n\t"
        "v_addc_co_u32 %[tot], %[m2], 0, %[tot], %[m2]\n\t"
        "s_lshl_b64 exec, exec, 1\n\t"
        ".set rc_i, rc_i + 2\n\t"
        ".endr\n\t"
        "s_mov_b64 exec, %[sav]\n\t"
        : [lo] "+v"(lo), [hi] "+v"(hi), [tot] "+v"(tot), [sk0] "=&s"(sk0), [sk1] "=&s"(sk1), [dd] "=&v"(dd),
          [m1] "=&s"(m1), [m2] "=&s"(m2), [sav] "=&s"(sav)
        : [key] "v"(key), [lowkey] "v"(lowkey), [c] "v"(c), [k256] "s"(256u)
        : "vcc");
}

// The modeler of ONE block, run by one wave (lane = symbol of the current tile of 64): the adaptive order-0 models of the block in
// LDS (two-level cumulative counts, rc_model.h; numeric models beyond the LDS slots in a global overflow area), and per tile every
// symbol's (cumLow, cumLow + freq, total) = the counts at the tile start + the earlier symbols of the tile, then
// Order0Model::update for the whole tile.  Used by the block coder below (records into an LDS ring for the coder wave) and by
// k_rc_records (records into global memory for host chains).
// CMP: a numeric group's byte-COUNT model (encodeNumeric's first symbol: 0..8 of its 256) keeps to the small layout in a place of its own
// -- Lw[x] = F(x) for x <= 16, total = F(16) + 240 -- so the slots, 280 words each, serve the byte models only: the nine or ten of them a
// read set uses all the time then fit the 12 a workgroup of 8 blocks has room for, and nothing hot is left to the global overflow area.
template <uint32_t RC_NSLOT, bool CMP>
struct RcModeler {
    static constexpr uint32_t SLOT_BASE = RC_SMALL_WORDS + (CMP ? RC_CMP_WORDS : 0);
    static constexpr uint32_t WORDS = SLOT_BASE + RC_NSLOT * RC_STRIDE;
    uint32_t* models; uint8_t* slotmap; uint32_t* gmodels; const uint16_t* sym16; int* err;
    uint64_t s0, s1;
    uint32_t nused, raw_next, small_sizes;
    // per-lane constants of the branch-free Order0Model::update of the overflow area: lanes 0..15 own Lw of the symbol's 16-block,
    // lanes 16..32 own H[0..16]
    bool is_lw; uint32_t upd_lane, upd_base, upd_blkmask;

    __device__ inline void init(uint32_t* models_, uint8_t* slotmap_, const uint16_t* sym16_, uint32_t small_sizes_, uint32_t lane, int* err_) {
        models = models_; slotmap = slotmap_; sym16 = sym16_; small_sizes = small_sizes_; err = err_;
        is_lw = lane < 16;
        upd_lane = is_lw ? lane : (lane < 33 ? lane - 16 : 0);
        upd_base = is_lw ? RC_LW + lane : upd_lane;
        upd_blkmask = is_lw ? ~0u : 0u;
        nused = 0; raw_next = 0xFFFFu; s0 = s1 = 0; gmodels = nullptr;
    }
    // AbstractDnaCoder::startBlock: the block's symbols are [s0_, s1_); the next tile's symbols are fetched one tile ahead
    // (a global load costs ~2000 cycles)
    __device__ inline void start_block(uint64_t s0_, uint64_t s1_, uint32_t* gmodels_, uint32_t lane) {
        s0 = s0_; s1 = s1_; gmodels = gmodels_; nused = 0;
        raw_next = lane < (uint32_t)((s1 - s0) < 64 ? (s1 - s0) : 64) ? sym16[s0 + lane] : 0xFFFFu;
        for (uint32_t m = 0; m < N_SMALL_MODELS + (CMP ? N_NUM_GROUPS : 0); m++) model_init(&models[m * RC_SSTRIDE], lane, true);
        for (uint32_t i = lane; i < RC_NNUM; i += 64) slotmap[i] = 255;
    }
    // tile t of the block: this lane's record (lanes past the block's end: cumLow 0, freq = total = 1, which leaves a chain as it is)
    __device__ inline void tile(uint32_t t, uint32_t lane, uint32_t& lo, uint32_t& hi, uint32_t& tot) {
        const uint64_t base = s0 + (uint64_t)t * 64;
        const uint32_t cnt = (uint32_t)((s1 - base) < 64 ? (s1 - base) : 64);
        const bool act = lane < cnt;
        const uint32_t raw = raw_next;
        {
            const uint64_t nb = base + 64;
            raw_next = (nb < s1 && lane < (uint32_t)((s1 - nb) < 64 ? (s1 - nb) : 64)) ? sym16[nb + lane] : 0xFFFFu;
        }
        const uint32_t m = raw & 0xff;
        const uint32_t grp = ((m - N_SMALL_MODELS) * 57u) >> 9;       // (m - 8) / 9 for the 72 numeric ids
        const bool cmodel = CMP && act && m >= N_SMALL_MODELS && (m - N_SMALL_MODELS) == grp * MODELS_PER_NUMERIC;
        uint32_t c = raw >> 8;
        if (CMP && __builtin_expect(__ballot(cmodel && c > 8u) != 0, 0)) {     // not from k_symbols (a value has at most 8 bytes): the launch fails, nothing is written out of place
            if (lane == 0) atomicExch(err, 3);
            if (cmodel && c > 8u) c = 8u;
        }
        const uint32_t key = (m << 8) | c;
        const bool numeric = act && m >= N_SMALL_MODELS && !cmodel;
        // slots for numeric models first seen in this tile
        uint32_t slot = numeric ? slotmap[m - N_SMALL_MODELS] : 0;
        unsigned long long need = __ballot(numeric && slot == 255);
        while (need) {
            const uint32_t l = (uint32_t)__builtin_ctzll(need);
            const uint32_t mm = (uint32_t)__builtin_amdgcn_readlane((int)m, (int)l);
            const uint32_t ns = nused++;
            if (lane == 0) slotmap[mm - N_SMALL_MODELS] = (uint8_t)ns;
            if (ns < RC_NSLOT) model_init(&models[SLOT_BASE + ns * RC_STRIDE], lane, false);
            else { model_init(gmodels + (uint64_t)(ns - RC_NSLOT) * RC_STRIDE, lane, false); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if (numeric && m == mm) slot = ns;
            need &= ~__ballot(numeric && m == mm);
        }
        // word offset of this lane's model (RC_GLOBAL | offset for the overflow area)
        uint32_t mb = 0;
        if (act) mb = cmodel ? RC_SMALL_WORDS + grp * RC_SSTRIDE
                    : !numeric ? m * RC_SSTRIDE
                               : (slot < RC_NSLOT ? SLOT_BASE + slot * RC_STRIDE : (RC_GLOBAL | ((slot - RC_NSLOT) * RC_STRIDE)));
        const uint32_t tot_idx = numeric ? 16u : RC_LW + (cmodel ? 16u : small_size_of(small_sizes, m));
        __builtin_amdgcn_wave_barrier();
        // counts at the tile start
        lo = 0; hi = 1; tot = 1;
        if (act) {
            if (!(mb & RC_GLOBAL)) {
                const uint32_t* s = &models[mb];
                lo = s[c >> 4] + s[RC_LW + c]; hi = s[(c + 1) >> 4] + s[RC_LW + c + 1]; tot = s[tot_idx] + (cmodel ? 240u : 0u);
            } else {
                const uint32_t* s = gmodels + (mb & ~RC_GLOBAL);
                lo = s[c >> 4] + s[RC_LW + c]; hi = s[(c + 1) >> 4] + s[RC_LW + c + 1]; tot = s[tot_idx];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // earlier symbols of the tile: lo += #(same model, smaller symbol), hi += #(same model, symbol <= mine),
        // tot += #(same model) over the lanes before this one -- 64 steps of a broadcast and three
        // compare/add-with-carry pairs; the lane mask "after lane i" is exec itself, shifted once per step
        // (one asm block, exec restored inside it)
        rc_rank_loop(key, c, lo, hi, tot);
        {
            if (!act) { lo = 0; hi = 1; tot = 1; }         // past the block's end: a record that leaves the chain as it is
        }
        // Order0Model::update for the whole tile, lane = symbol: F(x) += 1 for x > c, i.e. H[k] for k > c >> 4
        // and Lw[x] for the x after c inside c's 16-block (adds of 0 where it does not apply: no exec juggling)
        {
            // (only the lanes an add applies to take part: same-address adds serialise in the LDS, and with 8
            // modelers per CU its atomic unit is the busiest part of the kernel.  A small model's total is
            // read from Lw[size], so its H[] is never touched)
            const bool in_lds = act && !(mb & RC_GLOBAL);
            uint32_t* mp = &models[in_lds ? mb : 0];
            const uint32_t h4 = (in_lds && numeric) ? c >> 4 : 64u;
            const uint32_t l4 = in_lds ? c & 15u : 64u, ymax = numeric ? 15u : (cmodel ? 16u : small_size_of(small_sizes, m));
            uint32_t* lp = mp + RC_LW + (in_lds ? (c & ~15u) : 0u);
#pragma unroll
            for (uint32_t k = 1; k <= 16; k++)
                if (k > h4) (void)__hip_atomic_fetch_add(&mp[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
            for (uint32_t y = 1; y <= (CMP ? 16u : 15u); y++)
                if (y > l4 && y <= ymax) (void)__hip_atomic_fetch_add(&lp[y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            // symbols of models that live in the global overflow area: one at a time, lanes 0..32 own the entries
            unsigned long long gl = __ballot(act && (mb & RC_GLOBAL));
            while (gl) {
                const uint32_t i = (uint32_t)__builtin_ctzll(gl);
                gl &= gl - 1;
                const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)i);
                const uint32_t mbi = (uint32_t)__builtin_amdgcn_readlane((int)mb, (int)i);
                const uint32_t thr = is_lw ? (ci & 15u) : (ci >> 4);
                const uint32_t idx = upd_base + ((ci & ~15u) & upd_blkmask);
                if (upd_lane > thr)
                    (void)__hip_atomic_fetch_add(gmodels + (mbi & ~RC_GLOBAL) + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
};

// The coder's 64 steps of one tile, lane = block: RangeEncoder::encode for every record of the tile.  EXACT: the quotient by a
// 64-bit division (totals of 2^30 and more); else a truncated multiply-high by the total's reciprocal with a 32-bit fix-up.
template <bool EXACT>
__device__ inline void rc_coder_tile(const uint4* ra, const uint2* rb, uint64_t& low, uint64_t& range, uint32_t& nout, uint8_t* dst, uint32_t cap) {
    // the record TWO steps on is fetched under this step (one step ahead the coder still waited for the LDS now and then: 122.6 -> 118.9 ms
    // at 100 M reads, 175 -> 167 at the k = 63 shape; three ahead is slower again; the row's 65th entry is a spare)
    uint4 pa = ra[0], pa1 = ra[1];
    uint2 pb = rb[0], pb1 = rb[1];
#pragma clang loop unroll_count(EXACT ? 1 : 4)
    for (uint32_t j = 0; j < 64; j++) {
        const uint32_t s_lo = pa.x, s_fr = pa.y, s_tot = pa.z, s_hc = pa.w, b0 = pb.x, b1 = pb.y;
        const uint32_t jn = j + 2 < 64 ? j + 2 : 64;
        pa = pa1; pb = pb1; pa1 = ra[jn]; pb1 = rb[jn];
        // low += cumLow * q; range = q * freq; top = low + range = low + q * (cumLow + freq), q = floor(range / total)
        // (a third product and not `low + range`: the add waits for both products; profiles/r4_minimizer_filter.txt)
        uint64_t top;
        if (EXACT) {
            const uint64_t q = range / (uint64_t)s_tot;
            const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32);
            top = (uint64_t)q0 * s_hc + low;   top += (uint64_t)(q1 * s_hc) << 32;
            low = (uint64_t)q0 * s_lo + low;   low += (uint64_t)(q1 * s_lo) << 32;
            range = (uint64_t)q0 * s_fr;       range += (uint64_t)(q1 * s_fr) << 32;
        } else {
            // The step is as long as its DEPENDENT path (a lone wave issues in order; round 5: leaving instructions out of the step changed
            // nothing, an emitter wave's four and the modelers' updates alike -- DESIGN.md 4.3).  The quotient is a truncated multiply-high by the total's
            // reciprocal, at most 3 short (total < 2^30); the three products start from that estimate while the remainder is compared, and
            // the 0..3 it was short by goes into each with one multiply-add -- two levels fewer than fixing the quotient first.
            const uint32_t r0 = (uint32_t)range, r1 = (uint32_t)(range >> 32);
            uint64_t qe = (uint64_t)r1 * b1 + __umulhi(r1, b0);
            qe += __umulhi(r0, b1);
            const uint32_t q0 = (uint32_t)qe, q1 = (uint32_t)(qe >> 32);
            uint64_t T0 = (uint64_t)q0 * s_hc + low;   T0 += (uint64_t)(q1 * s_hc) << 32;
            uint64_t L0 = (uint64_t)q0 * s_lo + low;   L0 += (uint64_t)(q1 * s_lo) << 32;
            uint64_t R0 = (uint64_t)q0 * s_fr;         R0 += (uint64_t)(q1 * s_fr) << 32;
            asm volatile("" : "+v"(T0), "+v"(L0), "+v"(R0));        // (as written: the compiler would fold the fix-up back into the quotient)
            const uint32_t rem = r0 - q0 * s_tot;                   // true remainder < 4 * tot < 2^32
            const uint32_t t2 = s_tot << 1, t3 = t2 + s_tot;
            uint32_t e = (rem >= s_tot ? 1u : 0u) + (rem >= t2 ? 1u : 0u) + (rem >= t3 ? 1u : 0u);
            asm volatile("" : "+v"(e));
            top = (uint64_t)e * s_hc + T0;
            low = (uint64_t)e * s_lo + L0;
            range = (uint64_t)e * s_fr + R0;
        }
        // RangeEncoder::encode's while loop.  Usual case: low and low + range agree on their top 0..3
        // bytes and the shifted range stays >= BOTTOM: those bytes leave at once -- one unaligned 4-byte
        // store at the cursor (what lies past the cursor is overwritten by the stores that follow)
        const uint32_t lh = (uint32_t)(low >> 32);
        uint32_t xh = lh ^ (uint32_t)(top >> 32);
        uint32_t lead;                                              // v_ffbh_u32 answers -1 for 0: & 24 = 24 either way, without the `| 1` a level before it
        asm("v_ffbh_u32 %0, %1" : "=v"(lead), "+v"(xh));            // (xh in and out: the compiler would widen `xh == 0` below to a 64-bit compare of low ^ top)
        const uint32_t sh = lead & 24u;
        const uint64_t range_s = range << sh;
        // rare: four bytes or more leave, or the shifted range is below BOTTOM (one compare: the branch waits for it)
        uint32_t rare_below = xh == 0u ? 0xFFFFFFFFu : 0xFFFFu;
        asm volatile("" : "+v"(rare_below));
        const bool rare = (uint32_t)(range_s >> 32) <= rare_below;
        {
            const uint32_t cap4 = cap - 4;                          // (stores clamp to the block's room; the block reports overflow at its end)
            *(uint32_t*)(dst + (nout < cap4 ? nout : cap4)) = __builtin_bswap32(lh);
        }
        uint64_t low_n = low << sh, range_n = range_s;
        uint32_t nout_n = nout + (sh >> 3);
        // (the rare case is ONE scalar branch on "any lane", not an exec mask saved and restored around the lanes' own test)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0, 0) && rare) {
            // RangeEncoder::encode's loop as it is written
            low_n = low; range_n = range; nout_n = nout;
            while ((low_n ^ (low_n + range_n)) < (1ull << 56) ||
                   (range_n < RC_BOTTOM && ((range_n = (0 - low_n) & (RC_BOTTOM - 1)), true))) {
                if (nout_n < cap) dst[nout_n] = (uint8_t)(low_n >> 56);
                nout_n++;
                range_n <<= 8;
                low_n <<= 8;
            }
        }
        low = low_n; range = range_n; nout = nout_n;
    }
}

// One workgroup codes G blocks: G modeler waves, and one coder wave that runs the G serial chains
// in its lanes 0..G-1 (lane = block).  A chain step costs the same issue slots whether one lane or eight are active, so
// a CU that holds 8 blocks runs ONE chain instruction stream instead of eight; the step itself is branch-free in the
// usual cases (0..2 bytes leave, no range < BOTTOM reset) and falls back to RangeEncoder::encode's loop per lane otherwise.
// BIGOK: some block of the launch is long enough for a model's total to reach 2^30 (the host knows the blocks' symbol counts):
// that instantiation also carries the exact-division steps, tile by tile; the other one is the plain fast chain.
template <uint32_t G, uint32_t RC_NSLOT, bool BIGOK, bool CMP>
__global__ void __launch_bounds__(64 * rc_waves(G)) k_rc_encode(const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks,
                                                            uint8_t* out, const uint64_t* out_off, uint64_t* out_size,
                                                            uint32_t* scratch, int* err, uint32_t small_sizes, uint32_t fast_total) {
    constexpr uint32_t MW = RcModeler<RC_NSLOT, CMP>::WORDS;
    __shared__ uint32_t models_all[G * MW];
    // a step's record: {cumLow, freq, total, cumLow + freq} and the total's 64-bit reciprocal -- two LDS reads for the coder (its
    // step is paid in issue slots); one spare entry per row: the coder prefetches record j + 1
    __shared__ uint4 ring_a[G][2][RC_RING];
    __shared__ uint2 ring_b[G][2][RC_RING];
    __shared__ uint8_t slotmap_all[G][RC_NNUM];
    __shared__ uint32_t ntiles_s[G];
    __shared__ uint32_t tile_big[BIGOK ? G : 1][2];              // a total of fast_total or more in the tile: the coder divides exactly
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool is_coder = wave == 0, is_idle = wave != 0 && (wave & 3u) == 0;
    const uint32_t mi = is_coder || is_idle ? 0 : wave - 1 - (wave >> 2);     // the modeler's block inside the group
    RcModeler<RC_NSLOT, CMP> M;
    M.init(models_all + mi * MW, slotmap_all[mi], (const uint16_t*)syms, small_sizes, lane, err);

    const uint64_t n_groups = (n_blocks + G - 1) / G;
    for (uint64_t bg = blockIdx.x; bg < n_groups; bg += gridDim.x) {
        // this thread's block: the wave's for a modeler, the lane's for the coder
        const uint64_t b = bg * G + (is_coder ? lane : mi);
        bool valid = (is_coder ? lane < G : !is_idle) && b < n_blocks;
        uint64_t s0 = 0, s1 = 0;
        if (valid) { s0 = blk_begin[b]; s1 = blk_begin[b + 1]; }
        if (valid && s1 - s0 >= 0xFFFFFF00ull) {                 // the models' counts are 32-bit words (upstream's are 64-bit [RECALLED]: it would go on)
            if (is_coder || lane == 0) atomicExch(err, 2);
            if (is_coder) out_size[b] = 0;
            valid = false;
        }
        const uint32_t ntiles = valid ? (uint32_t)((s1 - s0 + 63) / 64) : 0;
        if (!is_coder && !is_idle && lane == 0) ntiles_s[mi] = ntiles;
        // coder state (wave 0, lane = block)
        uint64_t low = 0, range = ~0ull;
        uint32_t nout = 0;
        uint8_t* dst = nullptr;
        uint32_t cap = 0;                                        // stores stop at the block's room; the block reports overflow at its end (the cursor only grows)
        if (is_coder && valid) {
            dst = out + out_off[b];
            const uint64_t room = out_off[b + 1] - out_off[b];
            cap = (uint32_t)(room < 0xFFFFFFFFull ? room : 0xFFFFFFFFull);
        }
        if (!is_coder && !is_idle && valid)                        // AbstractDnaCoder::startBlock
            M.start_block(s0, s1, scratch + b * (uint64_t)(RC_NNUM - RC_NSLOT_SMALL) * RC_STRIDE, lane);
        __syncthreads();
        uint32_t T = 0;
        for (uint32_t w = 0; w < G; w++) T = ntiles_s[w] > T ? ntiles_s[w] : T;

        for (uint32_t t = 0; t <= T; t++) {
            if (is_idle) {
            } else if (!is_coder) {
                if (t < ntiles) {
                    // =================== modeler: tile t -> ring[mi][t & 1] ===================
                    uint32_t lo, hi, tot;
                    M.tile(t, lane, lo, hi, tot);
                    const uint64_t inv = ~0ull / (uint64_t)tot;        // per lane, off the serial chains
                    ring_a[mi][t & 1][lane] = make_uint4(lo, hi - lo, tot, hi);
                    ring_b[mi][t & 1][lane] = make_uint2((uint32_t)inv, (uint32_t)(inv >> 32));
                    if (BIGOK) {
                        const bool big = __ballot(tot >= fast_total) != 0;   // (the chain's 32-bit fix-up needs totals below 2^30)
                        if (lane == 0) tile_big[mi][t & 1] = big ? 1u : 0u;
                    }
                } else if (t < T) {
                    // the group's longer blocks go on: records that leave a chain as it is (cumLow 0, freq = total = 1)
                    ring_a[mi][t & 1][lane] = make_uint4(0u, 1u, 1u, 1u);
                    ring_b[mi][t & 1][lane] = make_uint2(~0u, ~0u);
                    if (BIGOK && lane == 0) tile_big[mi][t & 1] = 0u;
                }
            } else if (t > 0) {
                // =================== coder: tile t-1 of every block of the group, lane = block ===================
                // (symbols past a block's end are "leave as it is" records, so the 64 steps are uniform)
                if (valid) {
                    const uint4* ra = &ring_a[lane][(t - 1) & 1][0];
                    const uint2* rb = &ring_b[lane][(t - 1) & 1][0];
                    // (a tile in which some block's total has reached fast_total takes the same steps with the exact division)
                    bool any_big = false;
                    if (BIGOK) any_big = __ballot(tile_big[lane < G ? lane : 0][(t - 1) & 1] != 0 && lane < G) != 0;
                    if (!BIGOK || __builtin_expect(!any_big, 1)) rc_coder_tile<false>(ra, rb, low, range, nout, dst, cap);
                    else rc_coder_tile<true>(ra, rb, low, range, nout, dst, cap);
                }
            }
            __syncthreads();
        }
        if (is_coder && valid) {
            for (int i = 0; i < 8; i++) {                               // RangeEncoder::flush
                if (nout < cap) dst[nout] = (uint8_t)(low >> 56);
                nout++;
                low <<= 8;
            }
            out_size[b] = nout;
            if (nout > cap) atomicExch(err, 1);
        }
        __syncthreads();
    }
}

// ---- the modelers alone: the records of every symbol, for chains that run on HOST cores -------------------------------------------
// A launch of a few hundred blocks leaves most of the chip idle behind ONE block's serial chain per CU (126 ms for 1.16 M symbols:
// ~280 cycles per symbol on a lone wave), and a host core runs such a chain ~50 times faster.  So for small launches the device
// only does what it is good at -- the adaptive models' cumulative counts for every symbol, lane = symbol -- and writes a 64-bit
// record per symbol: cumLow | freq << 22 | model << 44 (the host keeps the models' totals itself: they are the symbols counted).
// The blocks go through in CHUNKS of tiles so that a chunk's records cross PCIe and are coded while the next chunk is modelled:
// a block's model state (its LDS image) waits in global memory between two launches.
constexpr uint32_t RCR_MW = RC_SMALL_WORDS + RC_NSLOT_BIG * RC_STRIDE;
constexpr uint32_t RCR_STATE_WORDS = RCR_MW + RC_NNUM / 4 + 4;     // models, slot map, number of slots used
size_t rc_records_state_bytes(uint64_t n_blocks) { return (size_t)n_blocks * RCR_STATE_WORDS * sizeof(uint32_t); }

__global__ void __launch_bounds__(64) k_rc_records(const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks, uint32_t tile0, uint32_t tile1,
                                                  uint64_t* recs, const uint64_t* rec_off, uint32_t* state, uint32_t* scratch, int* err, uint32_t small_sizes) {
    __shared__ uint32_t models[RCR_MW];
    __shared__ uint32_t slotmap_w[RC_NNUM / 4];
    uint8_t* slotmap = (uint8_t*)slotmap_w;
    const uint32_t lane = threadIdx.x;
    RcModeler<RC_NSLOT_BIG, false> M;
    M.init(models, slotmap, (const uint16_t*)syms, small_sizes, lane, err);
    for (uint64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const uint64_t s0 = blk_begin[b], s1 = blk_begin[b + 1];
        if (s1 - s0 >= (1ull << 22) - 512) { if (lane == 0) atomicExch(err, 2); continue; }     // (the host checks before it launches: 22-bit counts)
        const uint32_t ntiles = (uint32_t)((s1 - s0 + 63) / 64);
        const uint32_t ta = tile0 < ntiles ? tile0 : ntiles, tb = tile1 < ntiles ? tile1 : ntiles;
        if (ta >= tb) continue;
        uint32_t* st = state + b * (uint64_t)RCR_STATE_WORDS;
        uint32_t* gm = scratch + b * (uint64_t)(RC_NNUM - RC_NSLOT_SMALL) * RC_STRIDE;
        if (ta == 0) M.start_block(s0, s1, gm, lane);
        else {                                                   // the block goes on where the launch before left it
            for (uint32_t i = lane; i < RCR_MW; i += 64) models[i] = st[i];
            for (uint32_t i = lane; i < RC_NNUM / 4; i += 64) slotmap_w[i] = st[RCR_MW + i];
            M.s0 = s0; M.s1 = s1; M.gmodels = gm;
            M.nused = st[RCR_MW + RC_NNUM / 4];
            const uint64_t base = s0 + (uint64_t)ta * 64;
            M.raw_next = lane < (uint32_t)((s1 - base) < 64 ? (s1 - base) : 64) ? M.sym16[base + lane] : 0xFFFFu;
        }
        __syncthreads();
        uint64_t* out = recs + rec_off[b];
        for (uint32_t t = ta; t < tb; t++) {
            uint32_t lo, hi, tot;
            const uint32_t raw = M.raw_next;                     // (the tile's own symbols: tile() replaces them by the next tile's)
            M.tile(t, lane, lo, hi, tot);
            const uint64_t at = (uint64_t)(t - ta) * 64 + lane;
            if (s0 + (uint64_t)t * 64 + lane < s1) out[at] = (uint64_t)lo | ((uint64_t)(hi - lo) << 22) | ((uint64_t)(raw & 0xFFu) << 44);
        }
        __syncthreads();
        if (tb < ntiles) {                                       // to be continued
            for (uint32_t i = lane; i < RCR_MW; i += 64) st[i] = models[i];
            for (uint32_t i = lane; i < RC_NNUM / 4; i += 64) st[RCR_MW + i] = slotmap_w[i];
            if (lane == 0) st[RCR_MW + RC_NNUM / 4] = M.nused;
        }
        __syncthreads();
    }
}
// The same with FOUR waves per block (round 5).  A tile's 64-step rank loop -- nine tenths of the modeler's instructions -- depends on
// the tile's own symbols only, so wave w takes the loops of tiles w, w + 4, ... with counts that start from zero; what must go in tile
// order -- a new numeric model's slot, the counts at the tile start (added to the loop's), the record, Order0Model::update -- is a
// short serial section that the waves enter in turn (`turn` in LDS, raised behind the section's own LDS operations, which execute in
// order).  One block's pass: ~0.6 us per tile instead of a lone wave's 2.8.
__global__ void __launch_bounds__(256) k_rc_records4(const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks, uint32_t tile0, uint32_t tile1,
                                                    uint64_t* recs, const uint64_t* rec_off, uint32_t* state, uint32_t* scratch, int* err, uint32_t small_sizes) {
    __shared__ uint32_t models[RCR_MW];
    __shared__ uint32_t slotmap_w[RC_NNUM / 4];
    __shared__ uint32_t turn, nused_s;
    uint8_t* slotmap = (uint8_t*)slotmap_w;
    const uint16_t* sym16 = (const uint16_t*)syms;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr uint32_t NSLOT = RC_NSLOT_BIG;
    // per-lane constants of the overflow area's update (RcModeler::init)
    const bool is_lw = lane < 16;
    const uint32_t upd_lane = is_lw ? lane : (lane < 33 ? lane - 16 : 0), upd_base = is_lw ? RC_LW + lane : upd_lane, upd_blkmask = is_lw ? ~0u : 0u;
    for (uint64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const uint64_t s0 = blk_begin[b], s1 = blk_begin[b + 1];
        if (s1 - s0 >= (1ull << 22) - 512) { if (threadIdx.x == 0) atomicExch(err, 2); continue; }     // (the host checks before it launches: 22-bit counts)
        const uint32_t ntiles = (uint32_t)((s1 - s0 + 63) / 64);
        const uint32_t ta = tile0 < ntiles ? tile0 : ntiles, tb = tile1 < ntiles ? tile1 : ntiles;
        if (ta >= tb) continue;
        uint32_t* st = state + b * (uint64_t)RCR_STATE_WORDS;
        uint32_t* gm = scratch + b * (uint64_t)(RC_NNUM - RC_NSLOT_SMALL) * RC_STRIDE;
        __syncthreads();
        if (ta == 0) {                                           // AbstractDnaCoder::startBlock
            if (wv == 0) for (uint32_t m = 0; m < N_SMALL_MODELS; m++) model_init(&models[m * RC_SSTRIDE], lane, true);
            for (uint32_t i = threadIdx.x; i < RC_NNUM; i += 256) slotmap[i] = 255;
            if (threadIdx.x == 0) nused_s = 0;
        } else {                                                 // the block goes on where the launch before left it
            for (uint32_t i = threadIdx.x; i < RCR_MW; i += 256) models[i] = st[i];
            for (uint32_t i = threadIdx.x; i < RC_NNUM / 4; i += 256) slotmap_w[i] = st[RCR_MW + i];
            if (threadIdx.x == 0) nused_s = st[RCR_MW + RC_NNUM / 4];
        }
        if (threadIdx.x == 0) turn = ta;
        __syncthreads();
        uint64_t* out = recs + rec_off[b];
        auto load_raw = [&](uint32_t t) -> uint32_t {
            const uint64_t base = s0 + (uint64_t)t * 64;
            return (t < tb && lane < (uint32_t)((s1 - base) < 64 ? (s1 - base) : 64)) ? sym16[base + lane] : 0xFFFFu;
        };
        uint32_t raw_next = load_raw(ta + wv);
        for (uint32_t t = ta + wv; t < tb; t += 4) {
            const uint64_t base = s0 + (uint64_t)t * 64;
            const bool act = base + lane < s1;
            const uint32_t raw = raw_next;
            raw_next = load_raw(t + 4);                          // (a global load costs ~2000 cycles: the next tile's symbols one tile ahead)
            const uint32_t m = raw & 0xff, c = raw >> 8, key = (m << 8) | c;
            // ---- on its own: the tile's 64-step loop
            uint32_t lo = 0, hi = 0, tot = 0;
            rc_rank_loop(key, c, lo, hi, tot);
            // ---- in tile order
            while (__hip_atomic_load(&turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != t) __builtin_amdgcn_s_sleep(0);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            const bool numeric = act && m >= N_SMALL_MODELS;
            uint32_t slot = numeric ? slotmap[m - N_SMALL_MODELS] : 0;
            unsigned long long need = __ballot(numeric && slot == 255);
            if (need) {
                uint32_t nused = nused_s;
                while (need) {                                   // slots for numeric models first seen in this tile
                    const uint32_t l = (uint32_t)__builtin_ctzll(need);
                    const uint32_t mm = (uint32_t)__builtin_amdgcn_readlane((int)m, (int)l);
                    const uint32_t ns = nused++;
                    if (lane == 0) slotmap[mm - N_SMALL_MODELS] = (uint8_t)ns;
                    if (ns < NSLOT) model_init(&models[RC_SMALL_WORDS + ns * RC_STRIDE], lane, false);
                    else { model_init(gm + (uint64_t)(ns - NSLOT) * RC_STRIDE, lane, false); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    if (numeric && m == mm) slot = ns;
                    need &= ~__ballot(numeric && m == mm);
                }
                if (lane == 0) nused_s = nused;
            }
            uint32_t mb = 0;                                     // word offset of this lane's model (RC_GLOBAL | offset for the overflow area)
            if (act) mb = !numeric ? m * RC_SSTRIDE : (slot < NSLOT ? RC_SMALL_WORDS + slot * RC_STRIDE : (RC_GLOBAL | ((slot - NSLOT) * RC_STRIDE)));
            __builtin_amdgcn_wave_barrier();
            uint32_t blo = 0, bhi = 1;                           // the counts at the tile start (the records carry no totals: the host counts them)
            if (act) {
                if (!(mb & RC_GLOBAL)) {
                    const uint32_t* sm = &models[mb];
                    blo = sm[c >> 4] + sm[RC_LW + c]; bhi = sm[(c + 1) >> 4] + sm[RC_LW + c + 1];
                } else {                                         // (through L2: another wave's updates)
                    const uint32_t* sg = gm + (mb & ~RC_GLOBAL);
                    auto ld = [&](uint32_t i) { return __hip_atomic_load(sg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
                    blo = ld(c >> 4) + ld(RC_LW + c); bhi = ld((c + 1) >> 4) + ld(RC_LW + c + 1);
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (act) out[(uint64_t)(t - ta) * 64 + lane] = (uint64_t)(blo + lo) | ((uint64_t)((bhi + hi) - (blo + lo)) << 22) | ((uint64_t)(raw & 0xFFu) << 44);
            {   // Order0Model::update for the whole tile (RcModeler::tile's)
                const bool in_lds = act && !(mb & RC_GLOBAL);
                uint32_t* mp = &models[in_lds ? mb : 0];
                const uint32_t h4 = (in_lds && numeric) ? c >> 4 : 64u;
                const uint32_t l4 = in_lds ? c & 15u : 64u, ymax = numeric ? 15u : small_size_of(small_sizes, m);
                uint32_t* lp = mp + RC_LW + (in_lds ? (c & ~15u) : 0u);
#pragma unroll
                for (uint32_t k = 1; k <= 16; k++)
                    if (k > h4) (void)__hip_atomic_fetch_add(&mp[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (uint32_t y = 1; y <= 15; y++)
                    if (y > l4 && y <= ymax) (void)__hip_atomic_fetch_add(&lp[y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                unsigned long long gl = __ballot(act && (mb & RC_GLOBAL));
                while (gl) {
                    const uint32_t i = (uint32_t)__builtin_ctzll(gl);
                    gl &= gl - 1;
                    const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)i);
                    const uint32_t mbi = (uint32_t)__builtin_amdgcn_readlane((int)mb, (int)i);
                    const uint32_t thr = is_lw ? (ci & 15u) : (ci >> 4);
                    const uint32_t idx = upd_base + ((ci & ~15u) & upd_blkmask);
                    if (upd_lane > thr) (void)__hip_atomic_fetch_add(gm + (mbi & ~RC_GLOBAL) + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __hip_atomic_store(&turn, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        if (tb < ntiles) {                                       // to be continued
            for (uint32_t i = threadIdx.x; i < RCR_MW; i += 256) st[i] = models[i];
            for (uint32_t i = threadIdx.x; i < RC_NNUM / 4; i += 256) st[RCR_MW + i] = slotmap_w[i];
            if (threadIdx.x == 0) st[RCR_MW + RC_NNUM / 4] = nused_s;
        }
    }
}
void launch_rc_records(hipStream_t s, const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks, uint32_t tile0, uint32_t tile1, uint64_t* recs,
                       const uint64_t* rec_off, uint32_t* state, uint32_t* model_scratch, int* err, uint32_t small_sizes) {
    if (!n_blocks || tile0 >= tile1) return;
    static const bool one_wave = [] { const char* e = getenv("LEON_RC_RECORDS_WAVES"); return e && atoi(e) == 1; }();     // (measurement: the round-4 kernel)
    if (one_wave) hipLaunchKernelGGL(k_rc_records, dim3((uint32_t)std::min<uint64_t>(n_blocks, 256 * 4)), dim3(64), 0, s, syms, blk_begin, n_blocks, tile0, tile1, recs, rec_off,
                                     state, model_scratch, err, small_sizes);
    else hipLaunchKernelGGL(k_rc_records4, dim3((uint32_t)std::min<uint64_t>(n_blocks, 256 * 4)), dim3(256), 0, s, syms, blk_begin, n_blocks, tile0, tile1, recs, rec_off,
                            state, model_scratch, err, small_sizes);
}

void launch_rc_encode(hipStream_t s, const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks, uint8_t* out,
                      const uint64_t* out_off, uint64_t* out_size, uint32_t* model_scratch, int* err, uint64_t max_block_syms, uint32_t small_sizes, bool counts_apart) {
    if (!n_blocks) return;
    // blocks per workgroup: as few as keeps every block resident on the 256 CUs (LDS: 8 blocks of 19.6 KB per CU)
    const uint64_t per_cu = (n_blocks + 255) / 256;
    const char* force = getenv("LEON_RC_GROUP");              // measurement / test override: blocks per workgroup
    uint32_t G = per_cu <= 1 ? 1u : per_cu <= 2 ? 2u : per_cu <= 4 ? 4u : 8u;
    if (force) { int v = atoi(force); if (v == 1 || v == 2 || v == 4 || v == 8) G = (uint32_t)v; }
    const uint64_t n_groups = (n_blocks + G - 1) / G;
    const uint32_t g = (uint32_t)std::min<uint64_t>(n_groups, 256ull * (8 / G));
    // totals at which the coder switches to the exact division (test hook: a low value sends ordinary data down that path)
    uint32_t fast_total = RC_MAX_TOTAL;
    if (const char* e = getenv("LEON_RC_FAST_TOTAL_LOG2")) { const int v = atoi(e); if (v >= 4 && v <= 30) fast_total = 1u << v; }
    const bool big = max_block_syms + 512 >= fast_total;      // (a model's total is at most its block's symbol count + 256)
    // counts_apart (the read blocks' symbols, made by k_symbols: a numeric group's first model only ever sees 0..8): the byte-count models apart from
    // the 256-symbol slots (round 5); every other caller's streams -- the header stream's 14 byte models, leon_rc_encode_streams' arbitrary symbols --
    // as before.  LEON_RC_CMP=0: measurement / test, the round-4 layout for the read blocks too
    const bool cmp = counts_apart && [] { const char* e = getenv("LEON_RC_CMP"); return !e || atoi(e) != 0; }();
#define RC_LAUNCH(GG, N, B, C) hipLaunchKernelGGL((k_rc_encode<GG, N, B, C>), dim3(g), dim3(64 * rc_waves(GG)), 0, s, syms, blk_begin, n_blocks, out, out_off, out_size, model_scratch, err, small_sizes, fast_total)
#define RC_PICK(GG, N, C) do { if (big) RC_LAUNCH(GG, N, true, C); else RC_LAUNCH(GG, N, false, C); } while (0)
    if (cmp) {                                                // (slots at 8 blocks per workgroup: what 160 KB of LDS leave room for)
        if (G == 1) RC_PICK(1, RC_NSLOT_BIG, true);
        else if (G == 2) RC_PICK(2, RC_NSLOT_BIG, true);
        else if (G == 4) RC_PICK(4, RC_NSLOT_BIG, true);
        else RC_PICK(8, RC_NSLOT_SMALL, true);
    } else {
        if (G == 1) RC_PICK(1, RC_NSLOT_BIG, false);
        else if (G == 2) RC_PICK(2, RC_NSLOT_BIG, false);
        else if (G == 4) RC_PICK(4, RC_NSLOT_BIG, false);
        else RC_PICK(8, RC_NSLOT_SMALL + 1, false);
    }
#undef RC_PICK
#undef RC_LAUNCH
}

}  // namespace leon
