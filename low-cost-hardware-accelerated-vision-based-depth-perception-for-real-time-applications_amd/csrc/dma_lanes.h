// DMA lanes of the host-memory path: copies between page-locked host memory and HBM on SDMA engines WE choose.
//
// Why: hipMemcpyAsync picks "the first engine that is free right now" per copy (ROCm 7.2: rocblit.cpp logs copy_engine 0x1 / 0x2 /
// 0x4 wandering between a stream's copies), so a chunk's map download regularly lands on the engine the next chunk's image upload
// sits on and the two directions serialise - the host-to-host rate of round 4 moved between 20 400 and 26 100 pairs/s with the
// lottery (profiles/HISTORY.md, round 5).  The HSA runtime underneath takes the engine as an argument
// (hsa_amd_memory_async_copy_on_engine): uploads always on one engine, downloads always on another, each queue fed ahead of time.
//
// The HSA entry points are looked up in the runtime that is already loaded in the process (the one HIP itself runs on - PyTorch's
// wheel brings its own copy); when they are missing, or the GPU offers fewer than two engines towards the host, create() returns
// nullptr and the engine keeps using hipMemcpyAsync.  There is no reference counterpart: the reference copies synchronously, one
// pair at a time (src/parallel_includes/elas/elas_gpu.cu:537-563).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <string>

namespace sv {

class DmaLanes {
  public:
    enum Lane { UP = 0, DOWN = 1, DOWN2 = 2 };

    // device_ptr: any hipMalloc'd address of the GPU; host_ptr: any page-locked host address (they name the two agents).
    // lane_mask_override (experiments): bits 0-7 engine id of UP, 8-15 DOWN, 16-23 DOWN2 as log2 + 1 (0 = automatic).
    static DmaLanes *create(const void *device_ptr, const void *host_ptr, std::string *why, uint32_t lane_override = 0);
    ~DmaLanes();

    // A group of copies that complete together.  begin() -> ticket (>= 0) or -1; add() enqueues one copy (false: it could not be
    // enqueued - the group still completes for the copies that were); wait() blocks until all of them have landed and releases the
    // ticket (false: the runtime reported a failed copy).  nap: sleep between polls (throughput mode) instead of spinning.
    int begin(int ncopies);
    bool add(int ticket, Lane lane, void *dst, const void *src, size_t bytes, bool to_device);
    bool wait(int ticket, bool nap, int timeout_ms = 0);  // timeout_ms > 0: give up after that long (false; the ticket is NOT reused: its copies may still land)
    bool done(int ticket) const;  // all copies of the group have landed (does not release the ticket)

    uint32_t engine(Lane lane) const { return engine_[lane]; }
    std::string describe() const;

  private:
    DmaLanes() = default;
    struct Impl;
    Impl *impl_ = nullptr;
    uint32_t engine_[3] = {0, 0, 0};
};

}  // namespace sv
