// CPU emulation of csrc/delaunay_gpu.hip: the device functions compiled as plain C++, the tree processed depth by depth with
// the nodes of one depth in arbitrary (here: reversed) order, against Delaunay::triangulate.  tests/test_sanitizers.py builds it
// with AddressSanitizer; every loop of the merge gets a step budget so that a logic error ends the run instead of hanging.
#define DG_HOST_EMULATION 1
#define __device__
#define __forceinline__ inline
#include "delaunay_gpu.hip"
#include "host_stage.h"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

int main() {
    std::mt19937 rng(11);
    int bad = 0;
    for (int it = 0; it < 400; it++) {
        int n = 3 + rng() % (it % 10 == 0 ? 3900 : 400);
        std::vector<int32_t> xy(2 * n);
        for (int i = 0; i < n; i++) {
            if (it % 3 == 0) { xy[2 * i] = (int)(rng() % 250) * 5; xy[2 * i + 1] = (int)(rng() % 75) * 5; }
            else if (it % 3 == 1) { xy[2 * i] = (int)(rng() % 1300) - 50; xy[2 * i + 1] = (int)(rng() % 75) * 5; }
            else { xy[2 * i] = (int)(rng() % 12) * 5; xy[2 * i + 1] = (int)(rng() % 12) * 5; }
        }
        sv::Delaunay dl;
        std::vector<int32_t> want(6 * n + 24), ids(n);
        const int nw = dl.triangulate(xy.data(), n, want.data(), 2 * n + 8);
        sv::Delaunay dl2;
        const int m = dl2.kd_ordered_ids(xy.data(), n, ids.data());
        if (m < 3) { if (nw != 0) bad++; continue; }
        const int nslots = 2 * m - 1;
        std::vector<uint32_t> res(2 << 12, 0);
        std::vector<sv::dg::DTri> T(nslots);
        std::vector<int16_t> px(n), py(n);
        std::vector<uint16_t> ord(m);
        for (int i = 0; i < n; i++) { px[i] = (int16_t)xy[2 * i]; py[i] = (int16_t)xy[2 * i + 1]; }
        for (int i = 0; i < m; i++) ord[i] = (uint16_t)ids[i];
        memset((void *)&T[0], 0, sizeof(sv::dg::DTri));
        T[0].vtx[0] = T[0].vtx[1] = T[0].vtx[2] = 0xFFFF;
        const sv::dg::Mesh M{T.data(), px.data(), py.data()};
        for (int d = 12; d >= 0; d--)
            for (int j = (1 << d) - 1; j >= 0; j--) sv::dg::d_process_node(M, res.data(), ord.data(), m, d, j);
        std::vector<int32_t> got;
        for (int t = 1; t < nslots; t++) {
            if (T[t].vtx[0] == 0xFFFF || T[t].vtx[1] == 0xFFFF || T[t].vtx[2] == 0xFFFF) continue;
            got.push_back(T[t].vtx[1]); got.push_back(T[t].vtx[2]); got.push_back(T[t].vtx[0]);
        }
        if ((int)got.size() != 3 * nw || memcmp(got.data(), want.data(), sizeof(int32_t) * got.size())) {
            bad++;
            if (bad < 5) printf("mismatch: case %d n %d m %d  tris %d vs %zu\n", it, n, m, nw, got.size() / 3);
        }
    }
    printf("gpu-delaunay emulation done, mismatches: %d\n", bad);
    return bad != 0;
}
