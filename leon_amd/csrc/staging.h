// staging.h -- copies between PAGEABLE host memory and the device at PCIe's rate.  One hipMemcpy from or to pageable memory
// runs at 11-12 GB/s on this platform (the runtime stages it through one thread); a few threads that move 16 MiB pieces
// through pinned buffers, each on its own stream, reach 52-56 GB/s.  Used by the host entry points that take or
// return whole files' worth of bytes: leon_dna_encode_batch (capi.hip, with its own progress reporting), leon_dna_decode_blocks,
// the bloom and the leon_device_* helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace leon {

constexpr uint64_t kStagePiece = 16ull << 20;
constexpr uint32_t kStageThreads = 3;           // PCIe's rate already; more only take CPU time from whoever else is working

// the pinned buffers live as long as the process (allocating and pinning them costs tens of milliseconds)
struct StagePool {
    std::mutex mu;
    std::vector<void*> free_bufs;
    void* get() {
        { std::lock_guard<std::mutex> g(mu); if (!free_bufs.empty()) { void* p = free_bufs.back(); free_bufs.pop_back(); return p; } }
        void* p = nullptr;
        if (hipHostMalloc(&p, kStagePiece, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void put(void* p) { std::lock_guard<std::mutex> g(mu); free_bufs.push_back(p); }
};
inline StagePool& stage_pool() { static StagePool pool; return pool; }

inline uint32_t stage_threads() {
    static const uint32_t n = [] { const char* e = getenv("LEON_UPLOAD_THREADS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 32 ? (uint32_t)v : kStageThreads; }();
    return n;
}

// one worker's resources: a stream, two pinned buffers, an event per buffer
struct StageLane {
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    void* buf[2] = {nullptr, nullptr};
    bool ok = false;
    explicit StageLane(int dev) {
        buf[0] = stage_pool().get(); buf[1] = stage_pool().get();
        ok = buf[0] && buf[1] && hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
    }
    ~StageLane() {
        if (st) (void)hipStreamSynchronize(st);          // nothing of ours is in flight when the buffers go back
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (st) (void)hipStreamDestroy(st);
        for (void* b : buf) if (b) stage_pool().put(b);
    }
    StageLane(const StageLane&) = delete;
    StageLane& operator=(const StageLane&) = delete;
};

// device -> pageable host, blocking
inline hipError_t staged_d2h(int dev, void* dst, const void* d_src, uint64_t n) {
    if (!n) return hipSuccess;
    if (n < 4 * kStagePiece) return hipMemcpy(dst, d_src, n, hipMemcpyDeviceToHost);
    const uint64_t n_pieces = (n + kStagePiece - 1) / kStagePiece;
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&] {
        StageLane L(dev);
        if (!L.ok) { failed.store(1); return; }
        uint64_t held[2] = {~0ull, ~0ull};
        auto finish = [&](int t) -> bool {               // buffer t's piece has arrived: out to its place
            if (held[t] == ~0ull) return true;
            if (hipEventSynchronize(L.ev[t]) != hipSuccess) return false;
            const uint64_t a = held[t] * kStagePiece;
            memcpy((uint8_t*)dst + a, L.buf[t], std::min<uint64_t>(kStagePiece, n - a));
            held[t] = ~0ull;
            return true;
        };
        for (int turn = 0; !failed.load(); turn ^= 1) {
            if (!finish(turn)) { failed.store(1); break; }
            const uint64_t piece = next.fetch_add(1);
            if (piece >= n_pieces) break;
            const uint64_t a = piece * kStagePiece, m = std::min<uint64_t>(kStagePiece, n - a);
            if (hipMemcpyAsync(L.buf[turn], (const uint8_t*)d_src + a, m, hipMemcpyDeviceToHost, L.st) != hipSuccess ||
                hipEventRecord(L.ev[turn], L.st) != hipSuccess) { failed.store(1); break; }
            held[turn] = piece;
        }
        for (int t = 0; t < 2; t++) if (!failed.load() && !finish(t)) failed.store(1);
    };
    std::vector<std::thread> th;
    const uint32_t nt = (uint32_t)std::min<uint64_t>(stage_threads(), n_pieces);
    for (uint32_t i = 1; i < nt; i++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    (void)hipSetDevice(dev);
    return failed.load() ? hipErrorUnknown : hipSuccess;
}

// pageable host -> device, blocking
inline hipError_t staged_h2d(int dev, void* d_dst, const void* src, uint64_t n) {
    if (!n) return hipSuccess;
    if (n < 4 * kStagePiece) return hipMemcpy(d_dst, src, n, hipMemcpyHostToDevice);
    const uint64_t n_pieces = (n + kStagePiece - 1) / kStagePiece;
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&] {
        StageLane L(dev);
        if (!L.ok) { failed.store(1); return; }
        bool used[2] = {false, false};
        for (int turn = 0; !failed.load(); turn ^= 1) {
            if (used[turn] && hipEventSynchronize(L.ev[turn]) != hipSuccess) { failed.store(1); break; }   // the buffer's previous piece has left
            const uint64_t piece = next.fetch_add(1);
            if (piece >= n_pieces) break;
            const uint64_t a = piece * kStagePiece, m = std::min<uint64_t>(kStagePiece, n - a);
            memcpy(L.buf[turn], (const uint8_t*)src + a, m);      // (non-temporal stores instead: measured, no difference to the chain beside it)
            if (hipMemcpyAsync((uint8_t*)d_dst + a, L.buf[turn], m, hipMemcpyHostToDevice, L.st) != hipSuccess ||
                hipEventRecord(L.ev[turn], L.st) != hipSuccess) { failed.store(1); break; }
            used[turn] = true;
        }
        if (hipStreamSynchronize(L.st) != hipSuccess) failed.store(1);
    };
    std::vector<std::thread> th;
    const uint32_t nt = (uint32_t)std::min<uint64_t>(stage_threads(), n_pieces);
    for (uint32_t i = 1; i < nt; i++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    (void)hipSetDevice(dev);
    return failed.load() ? hipErrorUnknown : hipSuccess;
}

}  // namespace leon
