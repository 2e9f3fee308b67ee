#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include "context.hpp"

#include <string.h>

#include "kernels.hpp"

namespace ts {

Context::Context(int dev) : device(dev) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        throw Error(TS_ERR_HIP, "no HIP device available: the tap-stark HIP path has no CPU fallback");
    TS_REQUIRE(dev >= 0 && dev < count, TS_ERR_INVALID, "device index out of range");
    TS_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    TS_HIP(hipGetDeviceProperties(&prop, dev));
    num_cus = prop.multiProcessorCount;
    max_lds_per_block = prop.sharedMemPerBlock;  // gfx950: 160 KiB
    arch_name = prop.gcnArchName;
    arch_name = arch_name.substr(0, arch_name.find(':'));
    TS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
}

Context::~Context() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (fri_graph_exec) (void)hipGraphExecDestroy(fri_graph_exec);
    if (d_twiddle_fwd) (void)hipFree(d_twiddle_fwd);
    if (d_twiddle_inv) (void)hipFree(d_twiddle_inv);
    for (auto& t : scale_tables) (void)hipFree(t.d);
    for (auto& t : sel_tables)
        if (t.d) (void)hipFree(t.d);
    if (d_ticket) (void)hipFree(d_ticket);
    for (auto& kv : free_blocks) (void)hipFree(kv.second);
    for (auto& kv : live_blocks) (void)hipFree(kv.first);
    if (h_pinned) (void)hipHostFree(h_pinned);
    if (h_mailbox) (void)hipHostFree(h_mailbox);
    if (h_mailbox_big) (void)hipHostFree(h_mailbox_big);
    if (h_arena) (void)hipHostFree(h_arena);
    if (stream) (void)hipStreamDestroy(stream);
}

static size_t round_block(size_t bytes) {
    if (bytes < 512) return 512;
    if (bytes < (1u << 20)) {  // next power of two
        size_t r = 512;
        while (r < bytes) r <<= 1;
        return r;
    }
    const size_t g = 1u << 20;  // 1 MiB granules
    return (bytes + g - 1) / g * g;
}

void* Context::alloc(size_t bytes) {
    const size_t sz = round_block(bytes);
    auto it = free_blocks.find(sz);
    void* p = nullptr;
    if (alloc_log) alloc_log->push_back(sz);
    if (it != free_blocks.end()) {
        p = it->second;
        free_blocks.erase(it);
    } else {
        if (capturing) throw CaptureMiss{};  // hipMalloc / release_cache are illegal inside a capture
        TS_HIP(hipSetDevice(device));
        static const bool pool_debug = getenv("TS_POOL_DEBUG") != nullptr;
        if (pool_debug) {
            struct timespec ts_;
            clock_gettime(CLOCK_MONOTONIC, &ts_);
            fprintf(stderr, "[ts pool %p t=%.3f ms] hipMalloc %zu KiB (reserved %zu MiB)\n", (void*)this,
                    ts_.tv_sec * 1e3 + ts_.tv_nsec * 1e-6, sz >> 10, bytes_reserved >> 20);
        }
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess) {
            // drop the cache and retry once
            release_cache();
            e = hipMalloc(&p, sz);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();  // the refusal must not surface again at the next launch check
            throw Error(TS_ERR_OOM, "device allocation failed");
        }
        bytes_reserved += sz;
    }
    live_blocks[p] = sz;
    return p;
}

void Context::free(void* p) {
    if (!p) return;
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) return;
    if (capturing) {  // nothing captured has run: keep the block out of circulation until the capture ends
        deferred_free.push_back(p);
        return;
    }
    // All work is enqueued on the single stream, so a recycled block is only touched by kernels
    // that run after every earlier user of it.
    free_blocks.emplace(it->second, p);
    live_blocks.erase(it);
}

bool Context::reserve(const std::vector<size_t>& sizes) {
    std::map<size_t, size_t> need;
    for (size_t s : sizes) need[s]++;
    for (auto& kv : need) {
        for (size_t have = free_blocks.count(kv.first); have < kv.second; have++) {
            void* p = nullptr;
            (void)hipSetDevice(device);
            if (hipMalloc(&p, kv.first) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            bytes_reserved += kv.first;
            free_blocks.emplace(kv.first, p);
        }
    }
    return true;
}

std::vector<void*> Context::flush_deferred(const std::vector<void*>& revive) {
    std::vector<void*> revived;
    std::vector<void*> parked;
    parked.swap(deferred_free);
    for (void* p : parked) {
        bool keep = false;
        for (void* r : revive) keep = keep || r == p;
        if (keep) {
            bool seen = false;
            for (void* r : revived) seen = seen || r == p;
            if (!seen) revived.push_back(p);
            continue;
        }
        auto it = live_blocks.find(p);
        if (it == live_blocks.end()) continue;
        free_blocks.emplace(it->second, p);
        live_blocks.erase(it);
    }
    return revived;
}

void Context::release_cache() {
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto& kv : free_blocks) {
        (void)hipFree(kv.second);
        bytes_reserved -= kv.first;
    }
    free_blocks.clear();
}

uint32_t* Context::ticket() {
    if (!d_ticket) {
        TS_HIP(hipMalloc((void**)&d_ticket, 64 * 17));  // the global word + 16 group words, a line each
        TS_HIP(hipMemsetAsync(d_ticket, 0, 64 * 17, stream));
    }
    return d_ticket;
}

const void* Context::stage(const void* src, size_t bytes) {
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (need > h_arena_bytes / 2) {  // (re)allocate: rare, at most a few times per context
        if (capturing) throw CaptureMiss{};
        sync();
        if (h_arena) (void)hipHostFree(h_arena);
        h_arena = nullptr;
        size_t sz = 1 << 20;
        while (sz / 2 < need) sz <<= 1;
        TS_HIP(hipHostMalloc((void**)&h_arena, sz, hipHostMallocDefault));
        h_arena_bytes = sz;
        h_arena_off = 0;
    }
    if (h_arena_off + need > h_arena_bytes) {  // wrap: everything staged so far must have been consumed
        if (capturing) throw CaptureMiss{};
        sync();
        h_arena_off = 0;
    }
    void* p = h_arena + h_arena_off;
    memcpy(p, src, bytes);
    h_arena_off += need;
    return p;
}

void Context::sync_point(void* host, size_t bytes) {
    if (replay_mode == 2 && replay_pos < replay_log.size() && replay_log[replay_pos].size() == bytes) {
        memcpy(host, replay_log[replay_pos++].data(), bytes);
        return;
    }
    sync();
    if (replay_mode == 1) replay_log.emplace_back((const char*)host, (const char*)host + bytes);
}
void Context::d2h_point(void* host_dst, const void* dev_src, size_t bytes) {
    if (replay_mode == 2 && replay_pos < replay_log.size() && replay_log[replay_pos].size() == bytes) {
        memcpy(host_dst, replay_log[replay_pos++].data(), bytes);  // no copy enqueued: host_dst may die before it ran
        return;
    }
    if (bytes) TS_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, stream));
    sync();
    if (replay_mode == 1) replay_log.emplace_back((const char*)host_dst, (const char*)host_dst + bytes);
}

uint32_t* Context::mailbox(size_t words) {
    constexpr size_t WORDS = 16384;  // 64 KiB
    if (words > WORDS / 4) {
        // a message of its own size class (the opened values of a trace wider than ~500 columns): one
        // page-locked buffer grown on demand; its previous reader has synced, and growing it syncs again
        if (words > h_mailbox_big_words) {
            sync();
            if (h_mailbox_big) (void)hipHostFree(h_mailbox_big);
            h_mailbox_big = nullptr;
            h_mailbox_big_words = 0;
            TS_HIP(hipHostMalloc((void**)&h_mailbox_big, words * 4, hipHostMallocDefault));
            h_mailbox_big_words = words;
        }
        return h_mailbox_big;
    }
    if (!h_mailbox) TS_HIP(hipHostMalloc((void**)&h_mailbox, WORDS * 4, hipHostMallocDefault));
    const size_t need = (words + 15) & ~(size_t)15;
    if (mailbox_off + need > WORDS) mailbox_off = 0;  // every earlier slot has been read: its reader synced
    uint32_t* p = h_mailbox + mailbox_off;
    mailbox_off += need;
    return p;
}

void* Context::pinned(size_t bytes) {
    if (bytes > h_pinned_bytes) {
        if (h_pinned) {
            sync();
            (void)hipHostFree(h_pinned);
        }
        size_t sz = 1 << 16;
        while (sz < bytes) sz <<= 1;
        TS_HIP(hipHostMalloc(&h_pinned, sz, hipHostMallocDefault));
        h_pinned_bytes = sz;
    }
    return h_pinned;
}

std::map<std::string, std::pair<uint64_t, double>> Context::take_kernel_timings() {
    std::map<std::string, std::pair<uint64_t, double>> out;
    sync();
    for (auto& ev : kernel_events) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev.e0, ev.e1);
        auto& slot = out[ev.name];
        slot.first += 1;
        slot.second += ms;
        (void)hipEventDestroy(ev.e0);
        (void)hipEventDestroy(ev.e1);
    }
    kernel_events.clear();
    return out;
}

void Context::ensure_twiddles(unsigned log_size) {
    if (log_size <= twiddle_log && d_twiddle_fwd) return;
    TS_REQUIRE(log_size <= 27, TS_ERR_INVALID, "two-adicity of BabyBear is 27");
    sync();
    if (d_twiddle_fwd) (void)hipFree(d_twiddle_fwd);
    if (d_twiddle_inv) (void)hipFree(d_twiddle_inv);
    d_twiddle_fwd = d_twiddle_inv = nullptr;
    const size_t n = (size_t)1 << log_size;
    TS_HIP(hipMalloc((void**)&d_twiddle_fwd, n * 4));
    TS_HIP(hipMalloc((void**)&d_twiddle_inv, n * 4));
    twiddle_log = log_size;
    launch_build_twiddles(*this, d_twiddle_fwd, d_twiddle_inv, log_size);
}

}  // namespace ts
