// The dictionary chain with the renormalisation flag PREDICTED one symbol ahead (round 4), against the
// chain as it stands (host_rc.h encode_records), on records produced beforehand.  No GPU involved.
//   ./ab_spec [n_kmers] [skew]     skew: 0 uniform k-mers, 1 poly-A rich, 2 two-letter
//
// The carried recurrence of the chain as it stands is  q -> imul -> add -> xor -> cmp -> cmov -> q'  (7 cycles): the next
// quotient is selected by the flag f = "exactly one byte leaves", and f needs the step's new low and high ends.
// Here f(t + 1) is predicted during step t from step t's un-normalised (Lu, Ru) alone:
//     q(t+1) * lo'  ~  hi(Ru * G1'),   q(t+1) * (lo' + fr')  ~  hi(Ru * G2'),     G = floor(count * 2^32 / total') << 32
//     f(t+1)  ~  ((Lu + p1) ^ (Lu + pw)) < (f(t) ? 2^48 : 2^56)       (a byte that left at t moves the frame by 8 bits)
// so the recurrence becomes  q -> (shl) -> mulhi -> cmov -> q'  (6 cycles: the shifted quotient's product is one shift late),
// and the exact flag, computed as before but off the carried path, only CHECKS the prediction: a wrong one (the predicted
// ends are within ~2^32 of a byte boundary: ~10^-5 of the symbols) redoes the step's selections.  Bytes are identical by
// construction: every selection is finally made with the exact flag.
#define private public
#include "../../../leon_amd/csrc/host_rc.h"
#include <cstdio>
#include <cstring>
#include <random>
using namespace leon;

struct SpecRec { uint64_t c; uint32_t lo, g1, fr, g2; };     // 24 bytes: C = floor(fr 2^64 / (tot + 1)); g1 = floor(lo 2^32 / tot), g2 = floor((lo + fr) 2^32 / tot)

static constexpr uint64_t kTop = 1ull << 56, kBottom = 1ull << 48;

struct SpecCoder {
    uint64_t low_ = 0, range_ = ~0ull;
    std::vector<uint8_t> buf_;
    size_t w_ = 0;
    uint64_t mispredicted = 0, slow = 0, doubt = 0;       // (slow counts every plain step: rare cases, doubts and mispredictions)
    void settle() {
        uint8_t* p = buf_.data() + w_;
        while ((low_ ^ (low_ + range_)) < kTop || (range_ < kBottom && ((range_ = (0 - low_) & (kBottom - 1)), true))) {
            *p++ = (uint8_t)(low_ >> 56); range_ <<= 8; low_ <<= 8;
        }
        w_ = (size_t)(p - buf_.data());
    }
    void flush() {
        if (buf_.size() < w_ + 24) buf_.resize(w_ + 24);
        settle();
        for (int i = 0; i < 8; i++) { buf_[w_++] = (uint8_t)(low_ >> 56); low_ <<= 8; }
    }
    // One symbol the plain way, from the carried (q, L): used for whatever the fast path does not cover -- two bytes or more leaving at
    // once, the carry-less coder's range reset, a quotient in doubt, a mispredicted flag, a segment's last symbol (which must leave the
    // exact range behind).  Returns the new range; q, L, f, p are updated (f: the NEXT symbol's flag, exact, when there is a next record).
    struct Plain { uint64_t q, L, f, R; uint8_t* p; };
    __attribute__((noinline)) Plain step_plain(const SpecRec* r, bool has_next, uint64_t tot_next, uint64_t q, uint64_t L, uint64_t f, uint8_t* p) {
        uint64_t Lo = L + q * r->lo, R = q * r->fr;
        while ((Lo ^ (Lo + R)) < kTop || (R < kBottom && ((R = (0 - Lo) & (kBottom - 1)), true))) { *p++ = (uint8_t)(Lo >> 56); R <<= 8; Lo <<= 8; }
        const uint64_t qn = R / tot_next;
        if (has_next) f = ((Lo + qn * r[1].lo) ^ (Lo + qn * ((uint64_t)r[1].lo + r[1].fr))) < kTop ? 1 : 0;
        slow++;
        return Plain{qn, Lo, f, R, p};
    }
    // n symbols from their records, the first one being symbol t0 of the stream; r[n] must be readable (its g1, g2 feed the last
    // fast symbol's prediction)
    void encode_records(const SpecRec* r, uint64_t n, uint64_t t0) {
        if (!n) return;
        if (buf_.size() < w_ + 8 * n + 64) buf_.resize(buf_.size() * 2 + 8 * n + 4096);
        settle();
        uint8_t* p = buf_.data() + w_;
        uint64_t L = low_;
        const SpecRec* const r0 = r;
        uint64_t q = range_ / (5 + t0);
        // the first symbol's flag, exactly (the state is normalised here)
        uint64_t f = ((L + q * r->lo) ^ (L + q * ((uint64_t)r->lo + r->fr))) < kTop ? 1 : 0;
        const SpecRec* const e = r + n - 1;                      // the last symbol goes the plain way
        const uint64_t top = kTop, bottom = kBottom;
        while (r < e) {
            __builtin_prefetch(reinterpret_cast<const char*>(r) + 3072);
            const uint64_t lo = r->lo, fr = r->fr, C = r->c;
            const uint64_t Ru = q * fr;
            const uint64_t Lu = L + q * lo, x = Lu ^ (Lu + Ru);
            const uint64_t q8 = q << 8;
            const uint64_t Phi = (uint64_t)(((unsigned __int128)q * C) >> 64);
            const unsigned __int128 P8 = (unsigned __int128)q8 * C;
            const uint64_t Phi8 = (uint64_t)(P8 >> 64), Plo8 = (uint64_t)P8;
            // prediction of the NEXT symbol's flag from this symbol's un-normalised ends (the next record's g1 / g2 sit in the
            // high halves of two 8-byte words whose low halves -- lo, fr -- are noise far below the predictor's resolution)
            uint64_t G1, G2;
            memcpy(&G1, reinterpret_cast<const char*>(r + 1) + 8, 8);
            memcpy(&G2, reinterpret_cast<const char*>(r + 1) + 16, 8);
            const uint64_t xp = (Lu + (uint64_t)(((unsigned __int128)Ru * G1) >> 64)) ^ (Lu + (uint64_t)(((unsigned __int128)Ru * G2) >> 64));
            // whatever is not "no byte or exactly one, as predicted, with a quotient beyond doubt" goes the plain way, from (q, L)
            if (__builtin_expect((x < bottom) | (Ru < bottom) | ((x < top) != (f != 0)) | (Plo8 > ~q8), 0)) {
                mispredicted += (x >= bottom && Ru >= bottom && (x < top) != (f != 0));
                const Plain s = step_plain(r, true, 5 + t0 + (uint64_t)(r - r0) + 1, q, L, f, p);
                q = s.q; L = s.L; f = s.f; p = s.p;
                r++;
                continue;
            }
            *p = (uint8_t)(Lu >> 56);
            uint64_t qn = Phi, Ln = Lu, thr = top, fn;
            // (qn, Ln, thr) = f ? (Phi8, Lu << 8, 2^48) : (Phi, Lu, 2^56): conditional moves on the PREDICTED flag, known since the step before
            asm("testq %[f], %[f]\n\tcmovnzq %[A], %[qn]\n\tcmovnzq %[L1], %[Ln]\n\tcmovnzq %[B], %[thr]"
                : [qn] "+r"(qn), [Ln] "+r"(Ln), [thr] "+r"(thr)
                : [f] "r"(f), [A] "r"(Phi8), [L1] "r"(Lu << 8), [B] "r"(bottom)
                : "cc");
            p += f;
            asm("cmpq %[thr], %[xp]\n\tsbbq %[fn], %[fn]\n\tnegq %[fn]" : [fn] "=r"(fn) : [xp] "r"(xp), [thr] "r"(thr) : "cc");     // fn = xp < thr
            q = qn; L = Ln; f = fn; r++;
        }
        const Plain s = step_plain(r, false, 5 + t0 + n, q, L, f, p);
        slow--;
        range_ = s.R; low_ = s.L;
        w_ = (size_t)(s.p - buf_.data());
    }
    // The same, the fast path as ONE hand-scheduled asm loop (the compiler's schedule of the C form above spills the carried quotient
    // and puts its multiply last): state in registers, left at the first symbol that needs the plain way or at the end.
    // The predicted flag never lives in a register on the carried path: the compare that forms it (xp against thr, both carried from the
    // step before) sits right in front of the conditional moves that use it, so the path is  Ru -> mulhi -> add -> xor | cmp -> cmov.
    void encode_records_asm(const SpecRec* r, uint64_t n, uint64_t t0) {
        if (!n) return;
        if (buf_.size() < w_ + 8 * n + 64) buf_.resize(buf_.size() * 2 + 8 * n + 4096);
        settle();
        uint8_t* p = buf_.data() + w_;
        uint64_t L = low_;
        const SpecRec* const r0 = r;
        uint64_t q = range_ / (5 + t0);
        uint64_t f = ((L + q * r->lo) ^ (L + q * ((uint64_t)r->lo + r->fr))) < kTop ? 1 : 0;      // the first symbol's flag, exactly
        const SpecRec* const e = r + n - 1;                      // the last symbol goes the plain way
        while (r < e) {
            uint64_t xp = f ? 0 : 1, thr = 1;                    // (xp < thr) == f: the exact flag, in the form the loop carries its prediction
            asm volatile(
                ".p2align 5\n"
                "1:\n\t"
                "prefetcht0 3072(%[r])\n\t"
                "movq  %[q], %%rdx\n\t"
                "mulxq (%[r]), %%rbx, %%rbx\n\t"          // rbx = Phi = hi(q * C)
                "movl  8(%[r]), %%ecx\n\t"
                "imulq %[q], %%rcx\n\t"                   // q * lo
                "movl  16(%[r]), %%r13d\n\t"
                "imulq %[q], %%r13\n\t"                   // Ru = q * fr
                "shlq  $8, %%rdx\n\t"                     // q8
                "mulxq (%[r]), %%rax, %%r15\n\t"          // r15 = Phi8 = hi(q8 * C), rax = its low half
                "addq  %[L], %%rcx\n\t"                   // Lu
                "addq  %%rdx, %%rax\n\t"                  // quotient in doubt: the low half within q8 of 2^64
                "jc    2f\n\t"
                "rorxq $56, %%rcx, %%rdx\n\t"             // Lu rotated left by 8: dl = its top byte
                "movb  %%dl, (%[p])\n\t"
                "andq  $-256, %%rdx\n\t"                  // Lu << 8
                "cmpq  %[thr], %[xp]\n\t"                 // CF = this symbol's flag as the step before predicted it
                "cmovcq %%r15, %%rbx\n\t"                 // q' = f ? Phi8 : Phi
                "cmovncq %%rcx, %%rdx\n\t"                // L' = f ? Lu << 8 : Lu
                "movq  %[top], %[thr]\n\t"
                "cmovcq %[bot], %[thr]\n\t"               // the next symbol's byte boundary in this symbol's frame: f ? 2^48 : 2^56
                "sbbq  %%rax, %%rax\n\t"                  // the predicted flag as a mask
                "leaq  (%%rcx,%%r13), %%r15\n\t"
                "xorq  %%rcx, %%r15\n\t"                  // x = Lu ^ (Lu + Ru)
                "cmpq  %[bot], %%r15\n\t"
                "jb    2f\n\t"                            // two bytes or more
                "cmpq  %[bot], %%r13\n\t"
                "jb    2f\n\t"                            // range below BOTTOM
                "cmpq  %[top], %%r15\n\t"
                "sbbq  %%r15, %%r15\n\t"                  // the exact flag, as a mask
                "cmpq  %%rax, %%r15\n\t"
                "jne   2f\n\t"                            // mispredicted
                "subq  %%rax, %[p]\n\t"                   // p += f
                "movq  %%rbx, %[q]\n\t"
                "movq  %%rdx, %[L]\n\t"
                "movq  %%r13, %%rdx\n\t"
                "mulxq 32(%[r]), %%rax, %%rax\n\t"        // ~ q' * lo'
                "mulxq 40(%[r]), %%rdx, %%rdx\n\t"        // ~ q' * (lo' + fr')
                "addq  %%rcx, %%rax\n\t"
                "addq  %%rcx, %%rdx\n\t"
                "xorq  %%rdx, %%rax\n\t"
                "movq  %%rax, %[xp]\n\t"                  // the next symbol's predictor
                "addq  $24, %[r]\n\t"
                "cmpq  %[e], %[r]\n\t"
                "jb    1b\n\t"
                "movq  $-1, %%rax\n"                       // (left at the end: nothing to undo)
                "2:\n"
                : [r] "+r"(r), [q] "+r"(q), [L] "+r"(L), [p] "+r"(p), [xp] "+r"(xp), [thr] "+r"(thr)
                : [e] "m"(e), [top] "r"(kTop), [bot] "r"(kBottom)
                : "rax", "rbx", "rcx", "rdx", "r13", "r15", "cc", "memory");
            if (r >= e) { f = xp < thr ? 1 : 0; break; }
            // (left before anything of the symbol was committed: q, L, p are the step's inputs; its flag, whatever was predicted, is not needed)
            const Plain s = step_plain(r, true, 5 + t0 + (uint64_t)(r - r0) + 1, q, L, 0, p);
            q = s.q; L = s.L; f = s.f; p = s.p;
            r++;
        }
        const Plain s = step_plain(r, false, 5 + t0 + n, q, L, f, p);
        slow--;
        range_ = s.R; low_ = s.L;
        w_ = (size_t)(s.p - buf_.data());
    }
    // No prediction, as few micro-operations as the exact step allows (~31): for a core that sustains ~5 of them per cycle the step's
    // instruction count, not its carried latency, may be what bounds it.  Record: C, lo, lf = lo + fr (both ends' products start
    // together: q -> imul -> add -> xor -> cmp -> cmov, 7 cycles); Phi8 by a second multiply of the shifted quotient, whose low half
    // is the doubt test's operand as it comes.  `q8max` >= any q << 8 of the segment: doubt = low half > ~q8max (conservative).
    void encode_records_lean(const SpecRec* r, uint64_t n, uint64_t t0) {
        if (!n) return;
        if (buf_.size() < w_ + 8 * n + 64) buf_.resize(buf_.size() * 2 + 8 * n + 4096);
        settle();
        uint8_t* p = buf_.data() + w_;
        uint64_t L = low_;
        const SpecRec* const r0 = r;
        uint64_t q = range_ / (5 + t0);
        const SpecRec* const e = r + n - 1;
        // q <= (2^64 - 1) / total, total >= 5 + t0 > 256: q << 8 < 2^72 / (5 + t0)
        const uint64_t doubt_above = ~(uint64_t)((((unsigned __int128)1) << 72) / (5 + t0));
        while (r < e) {
            asm volatile(
                ".p2align 5\n"
                "1:\n\t"
                "prefetcht0 3072(%[r])\n\t"
                "movq  %[q], %%rdx\n\t"
                "mulxq (%[r]), %%rbx, %%rbx\n\t"          // rbx = Phi
                "movl  8(%[r]), %%ecx\n\t"
                "imulq %[q], %%rcx\n\t"                   // q * lo
                "movl  12(%[r]), %%r13d\n\t"
                "imulq %[q], %%r13\n\t"                   // q * lf
                "shlq  $8, %%rdx\n\t"
                "mulxq (%[r]), %%rax, %%r15\n\t"          // r15 = Phi8, rax = the low half of q8 * C
                "addq  %[L], %%rcx\n\t"                   // Lu
                "addq  %[L], %%r13\n\t"                   // Tu
                "cmpq  %[dbt], %%rax\n\t"
                "ja    2f\n\t"                            // quotient in doubt
                "movq  %%r13, %%rax\n\t"
                "xorq  %%rcx, %%rax\n\t"                  // x
                "subq  %%rcx, %%r13\n\t"                  // Ru
                "cmpq  %[bot], %%rax\n\t"
                "jb    2f\n\t"
                "cmpq  %[bot], %%r13\n\t"
                "jb    2f\n\t"
                "rorxq $56, %%rcx, %%rdx\n\t"
                "movb  %%dl, (%[p])\n\t"
                "andq  $-256, %%rdx\n\t"                  // Lu << 8
                "cmpq  %[top], %%rax\n\t"                 // CF = exactly one byte leaves
                "cmovcq %%r15, %%rbx\n\t"
                "cmovcq %%rdx, %%rcx\n\t"
                "adcq  $0, %[p]\n\t"
                "movq  %%rbx, %[q]\n\t"
                "movq  %%rcx, %[L]\n\t"
                "addq  $24, %[r]\n\t"
                "cmpq  %[e], %[r]\n\t"
                "jb    1b\n"
                "2:\n"
                : [r] "+r"(r), [q] "+r"(q), [L] "+r"(L), [p] "+r"(p)
                : [e] "m"(e), [top] "r"(kTop), [bot] "r"(kBottom), [dbt] "r"(doubt_above)
                : "rax", "rbx", "rcx", "rdx", "r13", "r15", "cc", "memory");
            if (r >= e) break;
            const Plain s = step_plain(r, false, 5 + t0 + (uint64_t)(r - r0) + 1, q, L, 0, p);
            q = s.q; L = s.L; p = s.p;
            r++;
        }
        const Plain s = step_plain(r, false, 5 + t0 + n, q, L, 0, p);
        slow--;
        range_ = s.R; low_ = s.L;
        w_ = (size_t)(s.p - buf_.data());
    }
    // LEAN without the redundant test and the prefetch, and unrolled by two so that nothing is moved between steps.
    // (x < 2^48 implies Ru < 2^48: adding Ru flips a bit of Lu at or above Ru's highest one, so x >= 2^floor(log2 Ru): the range test covers both.)
#define LEAN_STEP(OFF, QIN, LIN, QOUT, LOUT, LOUT32, EXIT)                                                            \
                "movq  " QIN ", %%rdx\n\t"                                                                             \
                "mulxq " OFF "(%[r]), " QOUT ", " QOUT "\n\t"                                                          \
                "movl  " OFF "+8(%[r]), " LOUT32 "\n\t"                                                                \
                "imulq " QIN ", " LOUT "\n\t"                                                                          \
                "movl  " OFF "+12(%[r]), %%r13d\n\t"                                                                   \
                "imulq " QIN ", %%r13\n\t"                                                                             \
                "shlq  $8, %%rdx\n\t"                                                                                  \
                "mulxq " OFF "(%[r]), %%rax, %%r15\n\t"                                                                \
                "addq  " LIN ", " LOUT "\n\t"                                                                          \
                "addq  " LIN ", %%r13\n\t"                                                                             \
                "cmpq  %[dbt], %%rax\n\t"                                                                              \
                "ja    " EXIT "\n\t"                                                                                   \
                "movq  %%r13, %%rax\n\t"                                                                               \
                "xorq  " LOUT ", %%rax\n\t"                                                                            \
                "subq  " LOUT ", %%r13\n\t"                                                                            \
                "cmpq  %[bot], %%r13\n\t"                                                                              \
                "jb    " EXIT "\n\t"                                                                                   \
                "rorxq $56, " LOUT ", %%rdx\n\t"                                                                       \
                "movb  %%dl, (%[p])\n\t"                                                                               \
                "andq  $-256, %%rdx\n\t"                                                                               \
                "cmpq  %[top], %%rax\n\t"                                                                              \
                "cmovcq %%r15, " QOUT "\n\t"                                                                           \
                "cmovcq %%rdx, " LOUT "\n\t"                                                                           \
                "adcq  $0, %[p]\n\t"
    void encode_records_lean2(const SpecRec* r, uint64_t n, uint64_t t0, bool unroll) {
        if (!n) return;
        if (buf_.size() < w_ + 8 * n + 64) buf_.resize(buf_.size() * 2 + 8 * n + 4096);
        settle();
        uint8_t* p = buf_.data() + w_;
        uint64_t L = low_;
        const SpecRec* const r0 = r;
        uint64_t q = range_ / (5 + t0);
        const SpecRec* const e = r + n - 1;
        const SpecRec* const e2 = e - 1;                          // the unrolled loop needs two records
        const uint64_t doubt_above = ~(uint64_t)((((unsigned __int128)1) << 72) / (5 + t0));
        while (r < e) {
            if (unroll && r < e2) {
                asm volatile(
                    ".p2align 5\n"
                    "1:\n\t"
                    LEAN_STEP("0", "%[q]", "%[L]", "%%rbx", "%%rcx", "%%ecx", "2f")
                    LEAN_STEP("24", "%%rbx", "%%rcx", "%[q]", "%[L]", "%k[L]", "3f")
                    "addq  $48, %[r]\n\t"
                    "cmpq  %[e], %[r]\n\t"
                    "jb    1b\n\t"
                    "jmp   2f\n"
                    "3:\n\t"                                   // left in the second half: the first half's results are the state, one record on
                    "movq  %%rbx, %[q]\n\t"
                    "movq  %%rcx, %[L]\n\t"
                    "addq  $24, %[r]\n"
                    "2:\n"
                    : [r] "+r"(r), [q] "+r"(q), [L] "+r"(L), [p] "+r"(p)
                    : [e] "m"(e2), [top] "r"(kTop), [bot] "r"(kBottom), [dbt] "r"(doubt_above)
                    : "rax", "rbx", "rcx", "rdx", "r13", "r15", "cc", "memory");
            } else if (!unroll) {
                asm volatile(
                    ".p2align 5\n"
                    "1:\n\t"
                    LEAN_STEP("0", "%[q]", "%[L]", "%%rbx", "%%rcx", "%%ecx", "2f")
                    "movq  %%rbx, %[q]\n\t"
                    "movq  %%rcx, %[L]\n\t"
                    "addq  $24, %[r]\n\t"
                    "cmpq  %[e], %[r]\n\t"
                    "jb    1b\n"
                    "2:\n"
                    : [r] "+r"(r), [q] "+r"(q), [L] "+r"(L), [p] "+r"(p)
                    : [e] "m"(e), [top] "r"(kTop), [bot] "r"(kBottom), [dbt] "r"(doubt_above)
                    : "rax", "rbx", "rcx", "rdx", "r13", "r15", "cc", "memory");
            }
            if (r >= e) break;
            // the record the loop stopped at (a rare case, or the odd record before the last): the plain way
            const Plain s = step_plain(r, false, 5 + t0 + (uint64_t)(r - r0) + 1, q, L, 0, p);
            q = s.q; L = s.L; p = s.p;
            r++;
        }
        const Plain s = step_plain(r, false, 5 + t0 + n, q, L, 0, p);
        slow--;
        range_ = s.R; low_ = s.L;
        w_ = (size_t)(s.p - buf_.data());
    }
    // LEAN2 x2 with NO register move on a carried path: the quotient lives in rdx (mulx's implicit operand), the shifted quotient is
    // formed by shlx into rax for a legacy mul whose high half lands in rdx -- where the next quotient belongs -- and the exclusive-or
    // is done in place.  (A mov that the renamer does not eliminate costs its cycle on the quotient's and on the flag's path.)
#define NOMOV_STEP(OFF, LIN, LOUT, LOUT32, EXIT)                                                                      \
                "movq  %%rdx, %%r14\n\t"                       /* the step's quotient, kept for a rare exit */           \
                "mulxq " OFF "(%[r]), %%rbx, %%rbx\n\t"        /* Phi */                                                 \
                "movl  " OFF "+8(%[r]), " LOUT32 "\n\t"                                                                \
                "imulq %%rdx, " LOUT "\n\t"                                                                            \
                "movl  " OFF "+12(%[r]), %%r13d\n\t"                                                                   \
                "imulq %%rdx, %%r13\n\t"                                                                               \
                "shlxq %[eight], %%rdx, %%rax\n\t"             /* q8 */                                                  \
                "mulq  " OFF "(%[r])\n\t"                      /* rdx = Phi8, rax = the low half */                      \
                "addq  " LIN ", " LOUT "\n\t"                  /* Lu */                                                  \
                "addq  " LIN ", %%r13\n\t"                     /* Tu */                                                  \
                "cmpq  %[dbt], %%rax\n\t"                                                                              \
                "ja    " EXIT "\n\t"                                                                                   \
                "movq  %%r13, %%rax\n\t"                                                                               \
                "subq  " LOUT ", %%rax\n\t"                    /* Ru */                                                  \
                "cmpq  %[bot], %%rax\n\t"                                                                              \
                "jb    " EXIT "\n\t"                                                                                   \
                "xorq  " LOUT ", %%r13\n\t"                    /* x, in place */                                         \
                "rorxq $56, " LOUT ", %%r15\n\t"                                                                       \
                "movb  %%r15b, (%[p])\n\t"                                                                             \
                "andq  $-256, %%r15\n\t"                       /* Lu << 8 */                                             \
                "cmpq  %[top], %%r13\n\t"                      /* CF = exactly one byte leaves */                        \
                "cmovncq %%rbx, %%rdx\n\t"                     /* q' = f ? Phi8 : Phi, in rdx */                         \
                "cmovcq %%r15, " LOUT "\n\t"                                                                           \
                "adcq  $0, %[p]\n\t"
    void encode_records_nomov(const SpecRec* r, uint64_t n, uint64_t t0) {
        if (!n) return;
        if (buf_.size() < w_ + 8 * n + 64) buf_.resize(buf_.size() * 2 + 8 * n + 4096);
        settle();
        uint8_t* p = buf_.data() + w_;
        uint64_t L = low_;
        const SpecRec* const r0 = r;
        uint64_t q = range_ / (5 + t0);
        const SpecRec* const e = r + n - 1;
        const SpecRec* const e2 = e - 1;
        const uint64_t doubt_above = ~(uint64_t)((((unsigned __int128)1) << 72) / (5 + t0));
        while (r < e) {
            if (r < e2) {
                asm volatile(
                    ".p2align 5\n"
                    "1:\n\t"
                    NOMOV_STEP("0", "%[L]", "%%rcx", "%%ecx", "2f")
                    NOMOV_STEP("24", "%%rcx", "%[L]", "%k[L]", "3f")
                    "addq  $48, %[r]\n\t"
                    "cmpq  %[e], %[r]\n\t"
                    "jb    1b\n\t"
                    "jmp   4f\n"
                    "3:\n\t"                                   // left in the second half: the first half's low end is the state, one record on
                    "movq  %%rcx, %[L]\n\t"
                    "addq  $24, %[r]\n"
                    "2:\n\t"
                    "movq  %%r14, %%rdx\n"                      // the quotient the step that was left began with
                    "4:\n"
                    : [r] "+r"(r), [q] "+d"(q), [L] "+r"(L), [p] "+r"(p)
                    : [e] "m"(e2), [top] "r"(kTop), [bot] "r"(kBottom), [dbt] "r"(doubt_above), [eight] "r"((uint64_t)8)
                    : "rax", "rbx", "rcx", "r13", "r14", "r15", "cc", "memory");
            }
            if (r >= e) break;
            const Plain s = step_plain(r, false, 5 + t0 + (uint64_t)(r - r0) + 1, q, L, 0, p);
            q = s.q; L = s.L; p = s.p;
            r++;
        }
        const Plain s = step_plain(r, false, 5 + t0 + n, q, L, 0, p);
        slow--;
        range_ = s.R; low_ = s.L;
        w_ = (size_t)(s.p - buf_.data());
    }
    // CARRY: the flag without the exclusive-or.  With everything shifted left by 8, "exactly one byte leaves" is "adding the range to the
    // low end's low 56 bits does not carry": a8 = (L << 8) + q * (lo << 8) = (Lu << 8) mod 2^64, r8 = q * (fr << 8) = (Ru << 8) mod 2^64, and the
    // carry of a8 + r8 is the flag -- multiply, add, add-with-carry, select: a 6-cycle path with no prediction.  A range of 2^56 or more
    // (no byte can leave; r8 has lost its top bits) is known from the quotient alone -- q >= ceil(2^56 / fr), a threshold the records carry --
    // and forces the carry by saturating r8.  Record: C, lo << 8, fr << 8, floor((2^56 - 1) / fr); quotient in rdx, unrolled by two.
    struct CarryRec { uint64_t c, lo8, fr8, qmm; };
#define CARRY_STEP(OFF, LIN, LOUT, LOUT32, EXIT)                                                                      \
                "movq  %%rdx, %%r14\n\t"                       /* the step's quotient, kept for a rare exit */           \
                "mulxq " OFF "(%[r]), %%rbx, %%rbx\n\t"        /* Phi */                                                 \
                "movq  " OFF "+8(%[r]), %%r12\n\t"                                                                     \
                "imulq %%rdx, %%r12\n\t"                       /* q * lo8 */                                             \
                "movq  " OFF "+16(%[r]), %%r13\n\t"                                                                    \
                "imulq %%rdx, %%r13\n\t"                       /* r8 = q * fr8 */                                        \
                "movl  " OFF "+9(%[r]), " LOUT32 "\n\t"        /* lo = bytes 1..4 of lo8 */                              \
                "imulq %%rdx, " LOUT "\n\t"                    /* q * lo */                                              \
                "cmpq  %%rdx, " OFF "+24(%[r])\n\t"            /* CF = floor((2^56 - 1) / fr) < q: the range is 2^56 or more */ \
                "sbbq  %%r15, %%r15\n\t"                                                                               \
                "shlxq %[eight], %%rdx, %%rax\n\t"             /* q8 */                                                  \
                "mulq  " OFF "(%[r])\n\t"                      /* rdx = Phi8, rax = the low half */                      \
                "cmpq  %[dbt], %%rax\n\t"                                                                              \
                "ja    " EXIT "\n\t"                                                                                   \
                "shlxq %[eight], " LIN ", %%rax\n\t"           /* L << 8 */                                              \
                "addq  %%rax, %%r12\n\t"                       /* a8 = (Lu << 8) mod 2^64 */                             \
                "addq  " LIN ", " LOUT "\n\t"                  /* Lu */                                                  \
                "orq   %%r15, %%r13\n\t"                       /* r8, saturated */                                       \
                "cmpq  %[top], %%r13\n\t"                                                                              \
                "jb    " EXIT "\n\t"                           /* range below 2^48 */                                    \
                "rorxq $56, " LOUT ", %%r15\n\t"                                                                       \
                "movb  %%r15b, (%[p])\n\t"                                                                             \
                "stc\n\t"                                                                                              \
                "adcq  %%r12, %%r13\n\t"                       /* CF = a byte boundary is crossed: NO byte leaves */     \
                "cmovcq %%rbx, %%rdx\n\t"                      /* q' = f ? Phi8 : Phi, in rdx */                         \
                "cmovncq %%r12, " LOUT "\n\t"                  /* L' = f ? Lu << 8 : Lu */                               \
                "sbbq  $-1, %[p]\n\t"                          /* p += f */
    void encode_records_carry(const CarryRec* r, const SpecRec* plain_recs, uint64_t n, uint64_t t0) {
        if (!n) return;
        if (buf_.size() < w_ + 8 * n + 64) buf_.resize(buf_.size() * 2 + 8 * n + 4096);
        settle();
        uint8_t* p = buf_.data() + w_;
        uint64_t L = low_;
        const CarryRec* const r0 = r;
        uint64_t q = range_ / (5 + t0);
        const CarryRec* const e = r + n - 1;
        const CarryRec* const e2 = e - 1;
        const uint64_t doubt_above = ~(uint64_t)((((unsigned __int128)1) << 72) / (5 + t0));
        while (r < e) {
            if (r < e2) {
                asm volatile(
                    ".p2align 5\n"
                    "1:\n\t"
                    CARRY_STEP("0", "%[L]", "%%rcx", "%%ecx", "2f")
                    CARRY_STEP("32", "%%rcx", "%[L]", "%k[L]", "3f")
                    "prefetcht0 2048(%[r])\n\t"                  /* (the records come from other cores' caches or from memory: a line per two symbols) */
                    "addq  $64, %[r]\n\t"
                    "cmpq  %[e], %[r]\n\t"
                    "jb    1b\n\t"
                    "jmp   4f\n"
                    "3:\n\t"
                    "movq  %%rcx, %[L]\n\t"
                    "addq  $32, %[r]\n"
                    "2:\n\t"
                    "movq  %%r14, %%rdx\n"
                    "4:\n"
                    : [r] "+r"(r), [q] "+d"(q), [L] "+r"(L), [p] "+r"(p)
                    : [e] "m"(e2), [top] "r"(kTop), [dbt] "r"(doubt_above), [eight] "r"((uint64_t)8)
                    : "rax", "rbx", "rcx", "r12", "r13", "r14", "r15", "cc", "memory");
            }
            if (r >= e) break;
            const Plain s = step_plain(plain_recs + (r - r0), false, 5 + t0 + (uint64_t)(r - r0) + 1, q, L, 0, p);
            q = s.q; L = s.L; p = s.p;
            r++;
        }
        const Plain s = step_plain(plain_recs + (r - r0), false, 5 + t0 + n, q, L, 0, p);
        slow--;
        range_ = s.R; low_ = s.L;
        w_ = (size_t)(s.p - buf_.data());
    }
};

int main(int argc, char** argv) {
    const uint32_t k = 31;
    const size_t n = argc > 1 ? atol(argv[1]) : 4000000;
    const int skew = argc > 2 ? atoi(argv[2]) : 0;
    std::mt19937_64 rng(1);
    std::vector<uint64_t> km(n);
    for (auto& x : km) {
        x = rng() >> 2;
        if (skew == 1) { const uint64_t m = rng() & rng() & rng(); x &= m >> 2; }           // mostly A
        if (skew == 2) x &= 0x5555555555555555ull >> 2;                                      // A and C only
    }
    std::vector<ChainRec16> recs(n * k);
    std::vector<SpecRec> srecs(n * k + 1);
    uint64_t c[4] = {1, 1, 1, 1}, t = 0;
    ChainRec16* out = recs.data();
    SpecRec* so = srecs.data();
    for (size_t a = 0; a < n; a++) for (uint32_t i = 0; i < k; i++, t++, out++, so++) {
        const uint32_t sy = (uint32_t)(km[a] >> (2 * (k - 1 - i))) & 3u;
        const uint64_t c01 = c[0] + c[1], lo = sy == 0 ? 0 : sy == 1 ? c[0] : sy == 2 ? c01 : c01 + c[2], fr = c[sy];
        uint64_t q, r; const uint64_t tot = 5 + t, d = tot + 1;
        asm("divq %[d]" : "=a"(q), "=d"(r) : "a"(0ull), "d"(fr), [d] "r"(d) : "cc");
        out->c = q; out->lo = (uint32_t)lo; out->fr = (uint32_t)fr;
        so->c = q; so->lo = (uint32_t)lo; so->fr = (uint32_t)fr;
        so->g1 = (uint32_t)((lo << 32) / tot); so->g2 = (uint32_t)(std::min<uint64_t>(((lo + fr) << 32) / tot, 0xFFFFFFFFull));
        c[sy]++;
    }
    srecs[n * k] = SpecRec{0, 0, 0, 0, 0};
    std::vector<SpecCoder::CarryRec> crecs(srecs.size());
    for (size_t i = 0; i < srecs.size(); i++)
        crecs[i] = SpecCoder::CarryRec{srecs[i].c, (uint64_t)srecs[i].lo << 8, (uint64_t)srecs[i].fr << 8, srecs[i].fr ? ((1ull << 56) - 1) / srecs[i].fr : 0};
    std::vector<SpecRec> lrecs(srecs);                              // the lean variant's records: {C, lo, lf = lo + fr, -, -} in the same 24 bytes
    for (auto& x : lrecs) x.g1 = x.lo + x.fr;
    const size_t plain = 20;
    // correctness: the whole stream both ways
    std::vector<uint8_t> ref;
    {
        AnchorDictCoder cd;
        cd.encode_kmers(km.data(), plain, k);
        auto t0 = std::chrono::steady_clock::now();
        cd.encode_records(recs.data() + plain * k, (n - plain) * k, plain * k);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        cd.flush();
        ref.assign(cd.data(), cd.data() + cd.size());
        printf("AS IT STANDS, records from DRAM: %.3f ns/symbol, %zu bytes\n", dt / ((n - plain) * (double)k) * 1e9, cd.size());
    }
    for (int rep = 0; rep < 14; rep++) {
        const bool use_asm = rep >= 2 && rep < 4, use_lean = rep >= 4 && rep < 6, use_lean2 = rep >= 6 && rep < 8, use_lean2x2 = rep >= 8 && rep < 10, use_nomov = rep >= 10 && rep < 12, use_carry = rep >= 12;
        AnchorDictCoder head;
        head.encode_kmers(km.data(), plain, k);
        SpecCoder sc;
        sc.buf_.assign(head.buf_.data(), head.buf_.data() + head.w_); sc.buf_.resize(sc.buf_.size() + 64);
        sc.w_ = head.w_; sc.low_ = head.low_; sc.range_ = head.range_;
        sc.buf_.resize(sc.buf_.size() + 8 * n * k + 4096);               // (room made and touched outside the timed region)
        auto t0 = std::chrono::steady_clock::now();
        for (uint64_t at = plain * k, end = n * k; at < end;) {         // segments of ~32 k symbols, as the library's feed hands them over
            const uint64_t m = std::min<uint64_t>(32767, end - at);
            if (use_carry) sc.encode_records_carry(crecs.data() + at, srecs.data() + at, m, at);
            else if (use_nomov) sc.encode_records_nomov(lrecs.data() + at, m, at);
            else if (use_lean2 || use_lean2x2) sc.encode_records_lean2(lrecs.data() + at, m, at, use_lean2x2);
            else if (use_lean) sc.encode_records_lean(lrecs.data() + at, m, at);
            else if (use_asm) sc.encode_records_asm(srecs.data() + at, m, at); else sc.encode_records(srecs.data() + at, m, at);
            at += m;
        }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        sc.flush();
        const bool same = sc.w_ == ref.size() && memcmp(sc.buf_.data(), ref.data(), ref.size()) == 0;
        printf("PREDICTED FLAG%s, records from DRAM: %.3f ns/symbol, %zu bytes, %s; mispredicted %llu, slow path %llu, in doubt %llu of %zu symbols\n",
               use_carry ? " -- none: CARRY x2 (asm)" : use_nomov ? " -- none: NO-MOV x2 (asm)" : use_lean2x2 ? " -- none: LEAN2 x2 (asm)" : use_lean2 ? " -- none: LEAN2 (asm)" : use_lean ? " -- none: LEAN (asm)" : use_asm ? " (asm)" : "", dt / ((n - plain) * (double)k) * 1e9, sc.w_, same ? "IDENTICAL" : "DIFFERENT", (unsigned long long)sc.mispredicted,
               (unsigned long long)sc.slow, (unsigned long long)sc.doubt, (n - plain) * (size_t)k);
        if (!same) return 1;
    }
    {   // cache-resident stretches, over and over (the state keeps evolving; output discarded by rewinding)
        const size_t m = 60000, start = (n > 1100000 ? 1000000 : n / 2) * (size_t)k;
        {
            AnchorDictCoder cd;
            cd.encode_kmers(km.data(), 20, k);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 300; i++) { cd.w_ = 0; cd.encode_records(recs.data() + start, m, start); }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("AS IT STANDS, records in cache: %.3f ns/symbol\n", dt / (300.0 * m) * 1e9);
        }
        {
            AnchorDictCoder head;
            head.encode_kmers(km.data(), 20, k);
            SpecCoder sc;
            sc.buf_.resize(1 << 20); sc.low_ = head.low_; sc.range_ = head.range_;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 300; i++) { sc.w_ = 0; sc.encode_records(srecs.data() + start, m, start); }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("PREDICTED FLAG, records in cache: %.3f ns/symbol (mispredicted %llu of %zu)\n", dt / (300.0 * m) * 1e9, (unsigned long long)sc.mispredicted, (size_t)300 * m);
        }
        {
            AnchorDictCoder head;
            head.encode_kmers(km.data(), 20, k);
            SpecCoder sc;
            sc.buf_.resize(1 << 20); sc.low_ = head.low_; sc.range_ = head.range_;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 300; i++) { sc.w_ = 0; sc.encode_records_asm(srecs.data() + start, m, start); }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("PREDICTED FLAG (asm), records in cache: %.3f ns/symbol (plain steps %llu of %zu)\n", dt / (300.0 * m) * 1e9, (unsigned long long)sc.slow, (size_t)300 * m);
        }
        {
            AnchorDictCoder head;
            head.encode_kmers(km.data(), 20, k);
            SpecCoder sc;
            sc.buf_.resize(1 << 20); sc.low_ = head.low_; sc.range_ = head.range_;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 300; i++) { sc.w_ = 0; sc.encode_records_lean(lrecs.data() + start, m, start); }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("NO PREDICTION, LEAN (asm), records in cache: %.3f ns/symbol (plain steps %llu of %zu)\n", dt / (300.0 * m) * 1e9, (unsigned long long)sc.slow, (size_t)300 * m);
        }
        for (int u = 0; u < 4; u++) {
            AnchorDictCoder head;
            head.encode_kmers(km.data(), 20, k);
            SpecCoder sc;
            sc.buf_.resize(1 << 20); sc.low_ = head.low_; sc.range_ = head.range_;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 300; i++) { sc.w_ = 0; if (u == 3) sc.encode_records_carry(crecs.data() + start, srecs.data() + start, m, start); else if (u == 2) sc.encode_records_nomov(lrecs.data() + start, m, start); else sc.encode_records_lean2(lrecs.data() + start, m, start, u == 1); }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("NO PREDICTION, %s%s (asm), records in cache: %.3f ns/symbol (plain steps %llu of %zu)\n", u == 3 ? "CARRY" : u == 2 ? "NO-MOV" : "LEAN2", u ? " x2" : "", dt / (300.0 * m) * 1e9, (unsigned long long)sc.slow, (size_t)300 * m);
        }
    }
    return 0;
}
