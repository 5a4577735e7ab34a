// Latency / throughput of the instructions the dictionary chain is made of, on the host it runs on (no GPU involved).
// Dependent chains give latency in ns; the add chain (1 cycle per step) gives the clock the core sustains, so everything
// can be read in cycles.  Independent streams give throughput (operations per cycle).
#include <chrono>
#include <cstdint>
#include <cstdio>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const uint64_t N = 400000000ull;
    uint64_t x = 12345, y = 0x9E3779B97F4A7C15ull;
    double t0, ns_add, ns;
    // dependent add: 1 cycle
    t0 = now();
    for (uint64_t i = 0; i < N; i += 8) {
        asm volatile("add %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0" : "+r"(x) : "r"(y));
    }
    ns_add = (now() - t0) / N * 1e9;
    printf("dependent add      : %.3f ns  -> clock %.2f GHz\n", ns_add, 1.0 / ns_add);
    // dependent imul r64, r64 (low half)
    t0 = now();
    for (uint64_t i = 0; i < N; i += 8) {
        asm volatile("imul %1, %0\n\timul %1, %0\n\timul %1, %0\n\timul %1, %0\n\timul %1, %0\n\timul %1, %0\n\timul %1, %0\n\timul %1, %0" : "+r"(x) : "r"(y));
    }
    ns = (now() - t0) / N * 1e9;
    printf("dependent imul     : %.3f ns = %.2f cycles\n", ns, ns / ns_add);
    // dependent mul r64 through the HIGH half: rax = x; mul y -> rdx:rax; x = rdx | 1<<62 (keeps it large): mul + or
    {
        uint64_t a = x | (1ull << 62), d = 0;
        t0 = now();
        for (uint64_t i = 0; i < N; i += 4) {
            asm volatile("mul %2\n\tmov %%rdx, %%rax\n\tmul %2\n\tmov %%rdx, %%rax\n\tmul %2\n\tmov %%rdx, %%rax\n\tmul %2\n\tmov %%rdx, %%rax"
                         : "+a"(a), "=&d"(d) : "r"(~0ull - 12345) : "cc");
        }
        ns = (now() - t0) / N * 1e9;
        printf("dependent mul (hi) : %.3f ns = %.2f cycles (incl. a mov, usually eliminated)\n", ns, ns / ns_add);
        x ^= a;
    }
    // dependent mulx through the high half
    {
        uint64_t a = x | (1ull << 62), lo;
        t0 = now();
        for (uint64_t i = 0; i < N; i += 4) {
            asm volatile("mulx %2, %1, %%rdx\n\tmulx %2, %1, %%rdx\n\tmulx %2, %1, %%rdx\n\tmulx %2, %1, %%rdx" : "+d"(a), "=&r"(lo) : "r"(~0ull - 12345) : "cc");
        }
        ns = (now() - t0) / N * 1e9;
        printf("dependent mulx (hi): %.3f ns = %.2f cycles\n", ns, ns / ns_add);
        x ^= a ^ lo;
    }
    // dependent cmov (through flags from a cmp on the value itself): cmp + cmov = 2 cycles if each is 1
    {
        uint64_t a = x, b = y;
        t0 = now();
        for (uint64_t i = 0; i < N; i += 4) {
            asm volatile("cmp %1, %0\n\tcmovb %1, %0\n\tcmp %1, %0\n\tcmovb %1, %0\n\tcmp %1, %0\n\tcmovb %1, %0\n\tcmp %1, %0\n\tcmovb %1, %0" : "+r"(a) : "r"(b) : "cc");
        }
        ns = (now() - t0) / N * 1e9;
        printf("dependent cmp+cmov : %.3f ns = %.2f cycles per pair\n", ns, ns / ns_add);
        x ^= a;
    }
    // throughput: independent imul (8 streams) and independent mulx-hi (6 streams)
    {
        uint64_t r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        t0 = now();
        for (uint64_t i = 0; i < N; i += 8) {
            asm volatile("imul %8, %0\n\timul %8, %1\n\timul %8, %2\n\timul %8, %3\n\timul %8, %4\n\timul %8, %5\n\timul %8, %6\n\timul %8, %7"
                         : "+r"(r[0]), "+r"(r[1]), "+r"(r[2]), "+r"(r[3]), "+r"(r[4]), "+r"(r[5]), "+r"(r[6]), "+r"(r[7]) : "r"(y));
        }
        ns = (now() - t0) / N * 1e9;
        printf("independent imul   : %.3f ns each = %.2f per cycle\n", ns, ns_add / ns);
        for (auto v : r) x ^= v;
    }
    {
        uint64_t h[6], src[6] = {11, 22, 33, 44, 55, 66};
        t0 = now();
        for (uint64_t i = 0; i < N; i += 6) {
            asm volatile("mulx %6, %0, %0\n\tmulx %7, %1, %1\n\tmulx %8, %2, %2\n\tmulx %9, %3, %3\n\tmulx %10, %4, %4\n\tmulx %11, %5, %5"
                         : "=&r"(h[0]), "=&r"(h[1]), "=&r"(h[2]), "=&r"(h[3]), "=&r"(h[4]), "=&r"(h[5])
                         : "r"(src[0]), "r"(src[1]), "r"(src[2]), "r"(src[3]), "r"(src[4]), "r"(src[5]), "d"(y));
        }
        ns = (now() - t0) / N * 1e9;
        printf("independent mulx   : %.3f ns each = %.2f per cycle\n", ns, ns_add / ns);
        for (int j = 0; j < 6; j++) x ^= h[j];
    }
    printf("(%llu)\n", (unsigned long long)x);
    return 0;
}
