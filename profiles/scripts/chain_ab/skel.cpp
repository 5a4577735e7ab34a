// Which part of the dictionary chain's step costs what: the LEAN step (ab_spec.cpp) taken apart.  Timing only -- the variants
// below drop tests and stores, so their bytes are not the coder's; the records are real ones (uniform k-mers).  No GPU involved.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
struct Rec { uint64_t c; uint32_t lo, lf, fr, pad; };
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static constexpr uint64_t kTop = 1ull << 56, kBottom = 1ull << 48;

#define PROLOGUE(OFF, QIN, QOUT)            "movq " QIN ", %%rdx\n\tmulxq " OFF "(%[r]), " QOUT ", " QOUT "\n\t"
#define ENDS(OFF, QIN, LIN, LOUT, LOUT32)   "movl " OFF "+8(%[r]), " LOUT32 "\n\timulq " QIN ", " LOUT "\n\tmovl " OFF "+12(%[r]), %%r13d\n\timulq " QIN ", %%r13\n\t" \
                                            "addq " LIN ", " LOUT "\n\taddq " LIN ", %%r13\n\t"
#define PHI8(OFF)                           "shlq $8, %%rdx\n\tmulxq " OFF "(%[r]), %%rax, %%r15\n\t"
#define DOUBT                               "cmpq %[dbt], %%rax\n\tja 2f\n\t"
#define XRU(LOUT)                           "movq %%r13, %%rax\n\txorq " LOUT ", %%rax\n\tsubq " LOUT ", %%r13\n\t"
#define RARE                                "cmpq %[bot], %%r13\n\tjb 2f\n\t"
#define BYTE(LOUT)                          "rorxq $56, " LOUT ", %%rdx\n\tmovb %%dl, (%[p])\n\tandq $-256, %%rdx\n\t"
#define SHIFTL(LOUT)                        "rorxq $56, " LOUT ", %%rdx\n\tandq $-256, %%rdx\n\t"
#define SELECT(QOUT, LOUT)                  "cmpq %[top], %%rax\n\tcmovcq %%r15, " QOUT "\n\tcmovcq %%rdx, " LOUT "\n\t"
#define PADV                                "adcq $0, %[p]\n\t"

#define LOOP2(BODY_A, BODY_B)                                                                                          \
    asm volatile(".p2align 5\n1:\n\t" BODY_A BODY_B "addq $48, %[r]\n\tcmpq %[e], %[r]\n\tjb 1b\n2:\n"                   \
                 : [r] "+r"(r), [q] "+r"(q), [L] "+r"(L), [p] "+r"(p)                                                   \
                 : [e] "m"(e), [top] "r"(kTop), [bot] "r"(kBottom), [dbt] "r"(dbt)                                      \
                 : "rax", "rbx", "rcx", "rdx", "r13", "r15", "cc", "memory");

#define APPLY(M, ...) M(__VA_ARGS__)
#define A_ "0", "%[q]", "%[L]", "%%rbx", "%%rcx", "%%ecx"
#define B_ "24", "%%rbx", "%%rcx", "%[q]", "%[L]", "%k[L]"
// step variants, each instantiated for both halves of the unrolled loop
#define FULL(OFF, QIN, LIN, QOUT, LOUT, L32)     PROLOGUE(OFF, QIN, QOUT) ENDS(OFF, QIN, LIN, LOUT, L32) PHI8(OFF) DOUBT XRU(LOUT) RARE BYTE(LOUT) SELECT(QOUT, LOUT) PADV
#define NOTESTS(OFF, QIN, LIN, QOUT, LOUT, L32)  PROLOGUE(OFF, QIN, QOUT) ENDS(OFF, QIN, LIN, LOUT, L32) PHI8(OFF) XRU(LOUT) BYTE(LOUT) SELECT(QOUT, LOUT) PADV
#define NOSTORE(OFF, QIN, LIN, QOUT, LOUT, L32)  PROLOGUE(OFF, QIN, QOUT) ENDS(OFF, QIN, LIN, LOUT, L32) PHI8(OFF) XRU(LOUT) SHIFTL(LOUT) SELECT(QOUT, LOUT)
#define NOPHI8(OFF, QIN, LIN, QOUT, LOUT, L32)   PROLOGUE(OFF, QIN, QOUT) "movq " QOUT ", %%r15\n\t" ENDS(OFF, QIN, LIN, LOUT, L32) XRU(LOUT) SHIFTL(LOUT) SELECT(QOUT, LOUT)
#define NOLSEL(OFF, QIN, LIN, QOUT, LOUT, L32)   PROLOGUE(OFF, QIN, QOUT) "movq " QOUT ", %%r15\n\t" ENDS(OFF, QIN, LIN, LOUT, L32) "movq %%r13, %%rax\n\txorq " LOUT ", %%rax\n\t" \
                                                 "cmpq %[top], %%rax\n\tcmovcq %%r15, " QOUT "\n\t"
// only the quotient's own recurrence: q -> mulx -> q'
#define QONLY(OFF, QIN, LIN, QOUT, LOUT, L32)    PROLOGUE(OFF, QIN, QOUT) "orq %[top], " QOUT "\n\t"

int main(int argc, char** argv) {
    const uint32_t k = 31;
    const size_t n = 2000000;
    std::mt19937_64 rng(1);
    std::vector<Rec> recs(n * k + 2);
    uint64_t c[4] = {1, 1, 1, 1}, t = 0;
    for (size_t a = 0; a < n; a++) {
        const uint64_t km = rng() >> 2;
        for (uint32_t i = 0; i < k; i++, t++) {
            const uint32_t sy = (uint32_t)(km >> (2 * (k - 1 - i))) & 3u;
            const uint64_t c01 = c[0] + c[1], lo = sy == 0 ? 0 : sy == 1 ? c[0] : sy == 2 ? c01 : c01 + c[2], fr = c[sy];
            const uint64_t d = 5 + t + 1;
            recs[t] = Rec{(uint64_t)((((unsigned __int128)fr) << 64) / d), (uint32_t)lo, (uint32_t)(lo + fr), (uint32_t)fr, 0};
            c[sy]++;
        }
    }
    const size_t m = 60000, start = 1000000 * (size_t)k;
    std::vector<uint8_t> out(1 << 20);
    const uint64_t dbt = ~(uint64_t)((((unsigned __int128)1) << 72) / (5 + start));
    const char* names[] = {"the whole step", "without the doubt and range tests", "... and without the byte store and cursor", "... and without the shifted quotient's multiply",
                           "... and without the low end's selection", "the quotient's recurrence alone (mulx)"};
    for (int rep = 0; rep < 2; rep++)
    for (int v = 0; v < 6; v++) {
        uint64_t q = (1ull << 60) / (5 + start), L = 12345;
        const double t0 = now();
        for (int it = 0; it < 300; it++) {
            const Rec* r = recs.data() + start;
            const Rec* e = r + m;
            uint8_t* p = out.data();
            if (q < (1ull << 20)) q |= 1ull << 28;                  // (keep the state in a plausible range when a variant lets it drift)
            while (r < e) {
                switch (v) {
                    case 0: LOOP2(APPLY(FULL, A_), APPLY(FULL, B_)) break;
                    case 1: LOOP2(APPLY(NOTESTS, A_), APPLY(NOTESTS, B_)) break;
                    case 2: LOOP2(APPLY(NOSTORE, A_), APPLY(NOSTORE, B_)) break;
                    case 3: LOOP2(APPLY(NOPHI8, A_), APPLY(NOPHI8, B_)) break;
                    case 4: LOOP2(APPLY(NOLSEL, A_), APPLY(NOLSEL, B_)) break;
                    case 5: LOOP2(APPLY(QONLY, A_), APPLY(QONLY, B_)) break;
                }
                if (r < e) { r += 2; q = (q >> 1) | (1ull << 30); }   // a rare exit: skip on (timing only)
            }
        }
        printf("%-52s %.3f ns per symbol\n", names[v], (now() - t0) / (300.0 * m) * 1e9);
    }
    return 0;
}
