// Probe of csrc/dma_lanes.cpp on the box at hand: which SDMA engines serve the host link, their rates alone and in pairs, and what
// sharing one engine between the directions costs (the lottery hipMemcpyAsync plays).
//   hipcc -O2 -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/sdma_lanes.cpp <pkg>/csrc/dma_lanes.cpp -I<pkg>/csrc -L/opt/rocm/lib -lamdhip64 -ldl -o abl_tmp/sdma_lanes
#include <hip/hip_runtime.h>
#include <string.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "dma_lanes.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

int main() {
    const size_t MB = 1 << 20, up = 60 * MB, dn = 119 * MB;
    const int reps = 8;
    char *d_up, *d_dn, *h_up, *h_dn;
    CK(hipMalloc(&d_up, up));
    CK(hipMalloc(&d_dn, dn));
    CK(hipHostMalloc(&h_up, up));
    CK(hipHostMalloc(&h_dn, dn));
    for (size_t i = 0; i < up; i++) h_up[i] = (char)(i * 7 + 3);
    CK(hipMemset(d_dn, 5, dn));
    CK(hipDeviceSynchronize());
    std::string why;
    sv::DmaLanes *def = sv::DmaLanes::create(d_up, h_up, &why);
    if (!def) {
        printf("no DMA lanes: %s\n", why.c_str());
        return 0;
    }
    printf("default: %s\n", def->describe().c_str());
    delete def;
    for (int eu = 1; eu <= 4; eu++)
        for (int ed = 1; ed <= 4; ed++) {
            sv::DmaLanes *l = sv::DmaLanes::create(d_up, h_up, &why, (uint32_t)eu | ((uint32_t)ed << 8) | ((uint32_t)ed << 16));
            if (!l) {  // (same engine for both directions is refused by create: probe it through DOWN2 = UP below)
                printf("engines up %d down %d: %s\n", eu, ed, why.c_str());
                continue;
            }
            double t_up = 0, t_dn = 0, t_both = 0;
            {
                const auto t0 = clk::now();
                const int t = l->begin(reps);
                for (int r = 0; r < reps; r++) l->add(t, sv::DmaLanes::UP, d_up, h_up, up, true);
                if (!l->wait(t, false)) printf("upload failed\n");
                t_up = ms_since(t0);
            }
            {
                const auto t0 = clk::now();
                const int t = l->begin(reps);
                for (int r = 0; r < reps; r++) l->add(t, sv::DmaLanes::DOWN, h_dn, d_dn, dn, false);
                if (!l->wait(t, false)) printf("download failed\n");
                t_dn = ms_since(t0);
            }
            {
                const auto t0 = clk::now();
                const int ta = l->begin(reps), tb = l->begin(reps);
                for (int r = 0; r < reps; r++) {
                    l->add(ta, sv::DmaLanes::UP, d_up, h_up, up, true);
                    l->add(tb, sv::DmaLanes::DOWN, h_dn, d_dn, dn, false);
                }
                l->wait(ta, false);
                l->wait(tb, false);
                t_both = ms_since(t0);
            }
            printf("engines up 0x%x down 0x%x: up alone %.1f GB/s, down alone %.1f GB/s, both %.1f + %.1f GB/s\n", l->engine(sv::DmaLanes::UP), l->engine(sv::DmaLanes::DOWN),
                   reps * up / t_up / 1e6, reps * dn / t_dn / 1e6, reps * up / t_both / 1e6, reps * dn / t_both / 1e6);
            fflush(stdout);
            delete l;
        }
    {  // both directions through ONE engine: uploads on UP, downloads on DOWN2 overridden to the same id
        sv::DmaLanes *l = sv::DmaLanes::create(d_up, h_up, &why, 1u | (2u << 8) | (1u << 16));
        if (l) {
            const auto t0 = clk::now();
            const int ta = l->begin(reps), tb = l->begin(reps);
            for (int r = 0; r < reps; r++) {
                l->add(ta, sv::DmaLanes::UP, d_up, h_up, up, true);
                l->add(tb, sv::DmaLanes::DOWN2, h_dn, d_dn, dn, false);
            }
            l->wait(ta, false);
            l->wait(tb, false);
            const double t = ms_since(t0);
            printf("ONE engine (0x%x) for both directions: %.1f + %.1f GB/s\n", l->engine(sv::DmaLanes::UP), reps * up / t / 1e6, reps * dn / t / 1e6);
            delete l;
        }
    }
    // the copies moved the bytes
    char *chk = (char *)malloc(up);
    CK(hipMemcpy(chk, d_up, up, hipMemcpyDeviceToHost));
    printf("upload intact: %d, download intact: %d\n", memcmp(chk, h_up, up) == 0, h_dn[0] == 5 && h_dn[dn - 1] == 5 && h_dn[dn / 2] == 5);
    // hipMemcpyAsync on two streams for comparison (the runtime's own engine choice)
    hipStream_t su, sd;
    CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    for (int pass = 0; pass < 2; pass++) {
        const auto t0 = clk::now();
        for (int r = 0; r < reps; r++) {
            CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, su));
            CK(hipMemcpyAsync(h_dn, d_dn, dn, hipMemcpyDeviceToHost, sd));
        }
        CK(hipDeviceSynchronize());
        const double t = ms_since(t0);
        if (pass) printf("hipMemcpyAsync, two streams: %.1f + %.1f GB/s\n", reps * up / t / 1e6, reps * dn / t / 1e6);
    }
    return 0;
}
