// DMA lanes (dma_lanes.h): engine-addressed SDMA copies through the HSA runtime HIP itself runs on.
#include "dma_lanes.h"

#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <time.h>

#include <mutex>
#include <vector>

namespace sv {

namespace {

struct HsaApi {
    decltype(&hsa_init) init = nullptr;
    decltype(&hsa_signal_create) signal_create = nullptr;
    decltype(&hsa_signal_destroy) signal_destroy = nullptr;
    decltype(&hsa_signal_load_scacquire) signal_load = nullptr;
    decltype(&hsa_signal_store_relaxed) signal_store = nullptr;
    decltype(&hsa_signal_subtract_screlease) signal_subtract = nullptr;
    decltype(&hsa_amd_pointer_info) pointer_info = nullptr;
    decltype(&hsa_amd_memory_async_copy_on_engine) copy_on_engine = nullptr;
    decltype(&hsa_amd_memory_copy_engine_status) engine_status = nullptr;
    decltype(&hsa_agent_get_info) agent_get_info = nullptr;
    bool ok = false;
};

// The HSA runtime that is already part of the process (the one HIP runs on; PyTorch's wheel brings its own copy under the same
// SONAME) - never a second instance: RTLD_NOLOAD only hands out what is loaded, whatever scope it was loaded into.
void *runtime_handle() {
    for (const char *name : {"libhsa-runtime64.so.1", "libhsa-runtime64.so"})
        if (void *h = dlopen(name, RTLD_NOLOAD | RTLD_NOW)) return h;
    return RTLD_DEFAULT;
}

template <class F>
bool bind(void *lib, F &f, const char *name) {
    f = reinterpret_cast<F>(dlsym(lib, name));
    return f != nullptr;
}

const HsaApi &api() {
    static HsaApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        void *lib = runtime_handle();
        bool ok = bind(lib, a.init, "hsa_init") && bind(lib, a.signal_create, "hsa_signal_create") && bind(lib, a.signal_destroy, "hsa_signal_destroy") &&
                  bind(lib, a.signal_load, "hsa_signal_load_scacquire") && bind(lib, a.signal_store, "hsa_signal_store_relaxed") &&
                  bind(lib, a.signal_subtract, "hsa_signal_subtract_screlease") && bind(lib, a.pointer_info, "hsa_amd_pointer_info") &&
                  bind(lib, a.copy_on_engine, "hsa_amd_memory_async_copy_on_engine") && bind(lib, a.engine_status, "hsa_amd_memory_copy_engine_status") &&
                  bind(lib, a.agent_get_info, "hsa_agent_get_info");
        // reference-counted by the runtime; HIP holds its own reference for the life of the process, ours is never given back
        a.ok = ok && a.init() == HSA_STATUS_SUCCESS;
    });
    return a;
}

bool agent_of(const HsaApi &a, const void *p, hsa_agent_t *out, hsa_device_type_t want) {
    hsa_amd_pointer_info_t info;
    info.size = sizeof(info);
    if (a.pointer_info(p, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    if (info.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return false;
    hsa_device_type_t type;
    if (a.agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != want) return false;
    *out = info.agentOwner;
    return true;
}

}  // namespace

struct DmaLanes::Impl {
    hsa_agent_t gpu{}, cpu{};
    std::mutex mu;
    std::vector<hsa_signal_t> signals;  // ticket = index
    std::vector<int> free_list;
    std::vector<char> failed;           // an add() of the group was refused by the runtime
};

DmaLanes *DmaLanes::create(const void *device_ptr, const void *host_ptr, std::string *why, uint32_t lane_override) {
    auto fail = [&](const char *msg) -> DmaLanes * {
        if (why) *why = msg;
        return nullptr;
    };
    const HsaApi &a = api();
    if (!a.ok) return fail("the HSA runtime of this process has no engine-addressed copies");
    Impl *im = new Impl();
    if (!agent_of(a, device_ptr, &im->gpu, HSA_DEVICE_TYPE_GPU) || !agent_of(a, host_ptr, &im->cpu, HSA_DEVICE_TYPE_CPU)) {
        delete im;
        return fail("hsa_amd_pointer_info does not know the staging buffers");
    }
    uint32_t up_mask = 0, down_mask = 0;
    if (a.engine_status(im->gpu, im->cpu, &up_mask) != HSA_STATUS_SUCCESS) up_mask = 0;      // dst = GPU: uploads
    if (a.engine_status(im->cpu, im->gpu, &down_mask) != HSA_STATUS_SUCCESS) down_mask = 0;  // dst = host: downloads
    DmaLanes *d = new DmaLanes();
    d->impl_ = im;
    auto lowest = [](uint32_t m) { return m & (0u - m); };
    // the host link is served at full rate by the first engines (the runtime itself only ever picked 0x1 / 0x2 / 0x4 for these copies)
    up_mask &= 0x7u;
    down_mask &= 0x7u;
    // downloads first: they are the larger direction (f32 maps: twice the bytes of the images); uploads take an engine they leave
    d->engine_[DOWN] = lowest(down_mask);
    d->engine_[UP] = lowest(up_mask & ~d->engine_[DOWN]);
    d->engine_[DOWN2] = lowest(down_mask & ~d->engine_[DOWN] & ~d->engine_[UP]);
    for (int l = 0; l < 3; l++) {
        const uint32_t o = (lane_override >> (8 * l)) & 0xFFu;
        if (o >= 1 && o <= 16) d->engine_[l] = 1u << (o - 1);
    }
    if (!d->engine_[UP] || !d->engine_[DOWN] || d->engine_[UP] == d->engine_[DOWN]) {
        delete d;
        return fail("fewer than two SDMA engines between the GPU and the host");
    }
    if (!d->engine_[DOWN2]) d->engine_[DOWN2] = d->engine_[DOWN];
    return d;
}

DmaLanes::~DmaLanes() {
    if (!impl_) return;
    const HsaApi &a = api();
    for (hsa_signal_t s : impl_->signals) (void)a.signal_destroy(s);
    delete impl_;
}

std::string DmaLanes::describe() const {
    char buf[96];
    snprintf(buf, sizeof(buf), "SDMA engines: uploads 0x%x, downloads 0x%x / 0x%x", engine_[UP], engine_[DOWN], engine_[DOWN2]);
    return buf;
}

int DmaLanes::begin(int ncopies) {
    if (ncopies <= 0) return -1;
    const HsaApi &a = api();
    std::lock_guard<std::mutex> lk(impl_->mu);
    int t;
    if (!impl_->free_list.empty()) {
        t = impl_->free_list.back();
        impl_->free_list.pop_back();
    } else {
        hsa_signal_t s;
        if (a.signal_create(0, 0, nullptr, &s) != HSA_STATUS_SUCCESS) return -1;
        impl_->signals.push_back(s);
        impl_->failed.push_back(0);
        t = (int)impl_->signals.size() - 1;
    }
    impl_->failed[t] = 0;
    a.signal_store(impl_->signals[t], ncopies);  // every finished copy takes one off
    return t;
}

bool DmaLanes::add(int ticket, Lane lane, void *dst, const void *src, size_t bytes, bool to_device) {
    const HsaApi &a = api();
    hsa_signal_t sig;
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        sig = impl_->signals[ticket];
    }
    hsa_status_t st = HSA_STATUS_SUCCESS;
    if (bytes > 0)
        st = a.copy_on_engine(dst, to_device ? impl_->gpu : impl_->cpu, src, to_device ? impl_->cpu : impl_->gpu, bytes, 0, nullptr, sig,
                              (hsa_amd_sdma_engine_id_t)engine_[lane], false);
    if (bytes == 0 || st != HSA_STATUS_SUCCESS) {
        a.signal_subtract(sig, 1);  // nothing in flight for this member of the group
        if (st != HSA_STATUS_SUCCESS) {
            std::lock_guard<std::mutex> lk(impl_->mu);
            impl_->failed[ticket] = 1;
            return false;
        }
    }
    return true;
}

bool DmaLanes::done(int ticket) const {
    const HsaApi &a = api();
    hsa_signal_t sig;
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        sig = impl_->signals[ticket];
    }
    return a.signal_load(sig) < 1;
}

bool DmaLanes::wait(int ticket, bool nap, int timeout_ms) {
    const HsaApi &a = api();
    hsa_signal_t sig;
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        sig = impl_->signals[ticket];
    }
    hsa_signal_value_t v;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while ((v = a.signal_load(sig)) >= 1) {
        if (timeout_ms > 0) {
            struct timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000L + (t1.tv_nsec - t0.tv_nsec) / 1000000L > timeout_ms) return false;  // (the ticket stays taken)
        }
        if (nap) {
            const struct timespec ts = {0, 30000};
            nanosleep(&ts, nullptr);
        } else {
            __builtin_ia32_pause();
        }
    }
    std::lock_guard<std::mutex> lk(impl_->mu);
    const bool ok = v == 0 && !impl_->failed[ticket];  // (the runtime stores a negative value into the signal of a failed copy)
    impl_->free_list.push_back(ticket);
    return ok;
}

}  // namespace sv
