// How many micro-operations per cycle the core the dictionary chain runs on really sustains (no GPU involved): the chain's loop is
// ~45 of them per symbol, and whether it is bound by its carried latency or by the core's width decides what can make it faster.
// Also: is the sibling hardware thread of the CPU we run on busy (another tenant)?  A busy sibling halves the front end.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <string>
#include <unistd.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool cpu_times(int cpu, unsigned long long& busy, unsigned long long& total) {
    FILE* f = fopen("/proc/stat", "r");
    if (!f) return false;
    char line[512]; char tag[32]; snprintf(tag, sizeof tag, "cpu%d ", cpu);
    bool ok = false;
    while (fgets(line, sizeof line, f)) {
        if (strncmp(line, tag, strlen(tag)) == 0) {
            unsigned long long v[8] = {0};
            sscanf(line + strlen(tag), "%llu %llu %llu %llu %llu %llu %llu %llu", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]);
            total = 0; for (auto x : v) total += x;
            busy = total - v[3] - v[4];
            ok = true; break;
        }
    }
    fclose(f);
    return ok;
}

int main() {
    const uint64_t N = 300000000ull;
    uint64_t a = 1, b = 2, c = 3, d = 4, e = 5, f = 6, g = 7, h = 8, k = 0x9E3779B97F4A7C15ull;
    double t0, ns_add;
    t0 = now();
    for (uint64_t i = 0; i < N; i += 8) asm volatile("add %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0" : "+r"(a) : "r"(k));
    ns_add = (now() - t0) / N * 1e9;
    printf("clock %.2f GHz (dependent adds)\n", 1.0 / ns_add);
    const int cpu = sched_getcpu();
    {   // the sibling hardware thread
        char path[128]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
        FILE* fs = fopen(path, "r"); char buf[64] = {0};
        if (fs) { if (fgets(buf, sizeof buf, fs)) {} fclose(fs); }
        printf("running on cpu %d, thread siblings: %s", cpu, buf[0] ? buf : "?\n");
    }
    // 8 independent add chains: adds per cycle
    t0 = now();
    for (uint64_t i = 0; i < N; i += 16)
        asm volatile("add %8, %0\n\tadd %8, %1\n\tadd %8, %2\n\tadd %8, %3\n\tadd %8, %4\n\tadd %8, %5\n\tadd %8, %6\n\tadd %8, %7\n\t"
                     "add %8, %0\n\tadd %8, %1\n\tadd %8, %2\n\tadd %8, %3\n\tadd %8, %4\n\tadd %8, %5\n\tadd %8, %6\n\tadd %8, %7"
                     : "+r"(a), "+r"(b), "+r"(c), "+r"(d), "+r"(e), "+r"(f), "+r"(g), "+r"(h) : "r"(k));
    double ns = (now() - t0) / N * 1e9;
    printf("independent adds        : %.2f per cycle\n", ns_add / ns);
    // adds + eliminated movs + a load and a store: the mix of a real loop
    {
        static uint64_t mem[64];
        t0 = now();
        for (uint64_t i = 0; i < N; i += 16)
            asm volatile("add %8, %0\n\tmov %0, %%rax\n\txor %8, %1\n\tmov (%9), %%rcx\n\tadd %%rcx, %2\n\tshl $3, %3\n\tadd %8, %4\n\tmov %4, %%rdx\n\t"
                         "sub %8, %5\n\tmovb %%al, 8(%9)\n\tadd %8, %6\n\txor %%rdx, %7\n\tadd %8, %0\n\tadd %8, %1\n\tcmp %8, %2\n\tcmovb %8, %3"
                         : "+r"(a), "+r"(b), "+r"(c), "+r"(d), "+r"(e), "+r"(f), "+r"(g), "+r"(h) : "r"(k), "r"(mem) : "rax", "rcx", "rdx", "cc", "memory");
        ns = (now() - t0) / N * 1e9;
        printf("mixed simple operations : %.2f per cycle (adds, xors, shifts, eliminated movs, a load, a store, cmp + cmov)\n", ns_add / ns);
    }
    // 2 imul + 4 mulx + 26 simple, all independent of one another across iterations except through their own registers
    {
        t0 = now();
        const uint64_t M = N / 2;
        for (uint64_t i = 0; i < M; i += 32)
            asm volatile("imul %8, %0\n\timul %8, %1\n\tmov %2, %%rdx\n\tmulx %8, %%rax, %%rcx\n\tmulx %3, %%rax, %%rcx\n\tmulx %4, %%rax, %%rcx\n\tmulx %5, %%rax, %%rcx\n\t"
                         "add %8, %2\n\tadd %8, %3\n\tadd %8, %4\n\tadd %8, %5\n\tadd %8, %6\n\tadd %8, %7\n\txor %8, %2\n\txor %8, %3\n\txor %8, %4\n\txor %8, %5\n\t"
                         "xor %8, %6\n\txor %8, %7\n\tsub %8, %2\n\tsub %8, %3\n\tsub %8, %4\n\tsub %8, %5\n\tsub %8, %6\n\tsub %8, %7\n\tadd %8, %2\n\tadd %8, %3\n\t"
                         "add %8, %4\n\tadd %8, %5\n\tadd %8, %6\n\tadd %8, %7\n\tadd %%rcx, %6"
                         : "+r"(a), "+r"(b), "+r"(c), "+r"(d), "+r"(e), "+r"(f), "+r"(g), "+r"(h) : "r"(k) : "rax", "rcx", "rdx", "cc");
        ns = (now() - t0) / M * 1e9;
        printf("2 imul + 4 mulx + 26 simple per 32 instructions: %.2f instructions per cycle\n", ns_add / ns);
    }
    unsigned long long b0 = 0, tt0 = 0, b1 = 0, tt1 = 0;
    // is the sibling busy?  (sampled over the runtime of one more add loop)
    int sib = -1;
    {
        char path[128]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
        FILE* fs = fopen(path, "r"); int x = -1, y = -1;
        if (fs) { if (fscanf(fs, "%d%*[,-]%d", &x, &y) >= 1) {} fclose(fs); }
        sib = x == cpu ? y : x;
    }
    if (sib >= 0 && cpu_times(sib, b0, tt0)) {
        t0 = now();
        while (now() - t0 < 1.0) for (uint64_t i = 0; i < 1000000; i += 8) asm volatile("add %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0\n\tadd %1, %0" : "+r"(a) : "r"(k));
        if (cpu_times(sib, b1, tt1) && tt1 > tt0) printf("sibling cpu %d was %.0f %% busy during one second of this run (as /proc/stat shows it)\n", sib, 100.0 * (b1 - b0) / (tt1 - tt0));
    }
    printf("(%llu)\n", (unsigned long long)(a ^ b ^ c ^ d ^ e ^ f ^ g ^ h));
    return 0;
}
