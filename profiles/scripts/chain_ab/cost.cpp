// What each instruction of the dictionary chain's loop costs the core in THROUGHPUT, measured in "add slots": a loop of 24
// independent adds (which the core retires at its full simple-ALU width) plus four instances of the instruction under test.
// cost = (cycles of the loop x adds per cycle of the plain loop - 24) / 4.  No GPU involved.
#include <chrono>
#include <cstdint>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static uint64_t mem[64];
#define ADDS6 "add %6,%0\n\tadd %6,%1\n\tadd %6,%2\n\tadd %6,%3\n\tadd %6,%4\n\tadd %6,%5\n\t"
#define ADDS24 ADDS6 ADDS6 ADDS6 ADDS6
#define RUN(name, X4)                                                                                                   \
    {                                                                                                                   \
        uint64_t a = 1, b = 2, c = 3, d = 4, e = 5, f = 6;                                                \
        const double t0 = now();                                                                                        \
        for (uint64_t i = 0; i < N; i++)                                                                                \
            asm volatile(ADDS24 X4 : "+r"(a), "+r"(b), "+r"(c), "+r"(d), "+r"(e), "+r"(f)              \
                         : "r"(k), "r"(mem) : "rax", "rbx", "rcx", "rdx", "r14", "cc", "memory");                 \
        const double cyc = (now() - t0) / N * 1e9 / ns_cycle;                                                           \
        if (base < 0) base = cyc;                                                                                       \
        printf("%-34s %6.2f cycles per iteration  -> %5.2f add-slots each\n", name, cyc, (cyc * 24.0 / base - 24.0) / 4.0); \
        sink ^= a ^ b ^ c ^ d ^ e ^ f;                                                                           \
    }
int main() {
    const uint64_t N = 40000000ull;
    uint64_t k = 0x9E3779B97F4A7C15ull, x = 1, sink = 0;
    double t0 = now();
    for (uint64_t i = 0; i < 8 * N; i += 8) asm volatile("add %1,%0\n\tadd %1,%0\n\tadd %1,%0\n\tadd %1,%0\n\tadd %1,%0\n\tadd %1,%0\n\tadd %1,%0\n\tadd %1,%0" : "+r"(x) : "r"(k));
    const double ns_cycle = (now() - t0) / (8 * N) * 1e9;
    printf("clock %.2f GHz\n", 1.0 / ns_cycle);
    double base = -1;
    RUN("24 adds alone", "")
    RUN("+ 4 nop-like (mov r,r)", "mov %0,%%rax\n\tmov %1,%%rbx\n\tmov %2,%%rcx\n\tmov %3,%%rdx\n\t")
    RUN("+ 4 imul r,r", "mov %6,%%rax\n\tmov %6,%%rbx\n\tmov %6,%%rcx\n\tmov %6,%%r14\n\timul %6,%%rax\n\timul %6,%%rbx\n\timul %6,%%rcx\n\timul %6,%%r14\n\t")
    RUN("  (the 4 movs of that alone)", "mov %6,%%rax\n\tmov %6,%%rbx\n\tmov %6,%%rcx\n\tmov %6,%%r14\n\t")
    RUN("+ 4 mulx r (hi only)", "mov %6,%%rdx\n\tmulx %0,%%rax,%%rax\n\tmulx %1,%%rbx,%%rbx\n\tmulx %2,%%rcx,%%rcx\n\tmulx %3,%%r14,%%r14\n\t")
    RUN("+ 4 mulx r (hi and lo)", "mov %6,%%rdx\n\tmulx %0,%%rax,%%rbx\n\tmulx %1,%%rcx,%%r14\n\tmulx %2,%%rax,%%rbx\n\tmulx %3,%%rcx,%%r14\n\t")
    RUN("+ 4 mulx m (hi and lo)", "mov %6,%%rdx\n\tmulx (%7),%%rax,%%rbx\n\tmulx 8(%7),%%rcx,%%r14\n\tmulx 16(%7),%%rax,%%rbx\n\tmulx 24(%7),%%rcx,%%r14\n\t")
    RUN("+ 4 mul r (rdx:rax)", "mov %6,%%rax\n\tmul %0\n\tmov %6,%%rax\n\tmul %1\n\tmov %6,%%rax\n\tmul %2\n\tmov %6,%%rax\n\tmul %3\n\t")
    RUN("+ 4 shld $8", "shld $8,%0,%%rax\n\tshld $8,%1,%%rbx\n\tshld $8,%2,%%rcx\n\tshld $8,%3,%%r14\n\t")
    RUN("+ 4 rorx", "rorx $56,%0,%%rax\n\trorx $56,%1,%%rbx\n\trorx $56,%2,%%rcx\n\trorx $56,%3,%%r14\n\t")
    RUN("+ 4 shl $8", "mov %0,%%rax\n\tshl $8,%%rax\n\tmov %1,%%rbx\n\tshl $8,%%rbx\n\tmov %2,%%rcx\n\tshl $8,%%rcx\n\tmov %3,%%r14\n\tshl $8,%%r14\n\t")
    RUN("+ 4 (cmp + cmov)", "cmp %6,%0\n\tcmovb %1,%%rax\n\tcmp %6,%1\n\tcmovb %2,%%rbx\n\tcmp %6,%2\n\tcmovb %3,%%rcx\n\tcmp %6,%3\n\tcmovb %4,%%r14\n\t")
    RUN("+ 4 cmp alone", "cmp %6,%0\n\tcmp %6,%1\n\tcmp %6,%2\n\tcmp %6,%3\n\t")
    RUN("+ 4 (cmp + jb not taken)", "cmp %0,%0\n\tjb 9f\n\tcmp %1,%1\n\tjb 9f\n\tcmp %2,%2\n\tjb 9f\n\tcmp %3,%3\n\tjb 9f\n\t9:\n\t")
    RUN("+ 4 adc $0", "adc $0,%%rax\n\tadc $0,%%rbx\n\tadc $0,%%rcx\n\tadc $0,%%r14\n\t")
    RUN("+ 4 sbb r,r", "sbb %%rax,%%rax\n\tsbb %%rbx,%%rbx\n\tsbb %%rcx,%%rcx\n\tsbb %%r14,%%r14\n\t")
    RUN("+ 4 loads (8 bytes)", "mov (%7),%%rax\n\tmov 8(%7),%%rbx\n\tmov 16(%7),%%rcx\n\tmov 24(%7),%%r14\n\t")
    RUN("+ 4 byte stores", "movb %%al,64(%7)\n\tmovb %%bl,65(%7)\n\tmovb %%cl,66(%7)\n\tmovb %%dl,67(%7)\n\t")
    RUN("+ 4 prefetcht0", "prefetcht0 128(%7)\n\tprefetcht0 192(%7)\n\tprefetcht0 256(%7)\n\tprefetcht0 320(%7)\n\t")
    RUN("+ 4 lea (r,r)", "lea (%0,%1),%%rax\n\tlea (%1,%2),%%rbx\n\tlea (%2,%3),%%rcx\n\tlea (%3,%4),%%r14\n\t")
    RUN("+ 4 xor r,r'", "mov %0,%%rax\n\txor %1,%%rax\n\tmov %1,%%rbx\n\txor %2,%%rbx\n\tmov %2,%%rcx\n\txor %3,%%rcx\n\tmov %3,%%r14\n\txor %4,%%r14\n\t")
    printf("(%llu)\n", (unsigned long long)(sink ^ x));
    return 0;
}
