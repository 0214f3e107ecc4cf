/*
 * TEST INFRASTRUCTURE — CPU restatement of the reference's ELAS hot path (see elas_oracle.h).
 *
 * Plain scalar C++, IEEE arithmetic (build with -ffp-contract=off, never -ffast-math), every scratch
 * buffer zero-filled (the canonical oracle state of SURVEY.md §0 fact 5).  Each function cites the
 * reference lines it follows; paths are relative to /root/reference/src.
 */
#include "elas_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <vector>

namespace {

/* ------------------------------------------------------------------------------------------------
 * Stage 1: Sobel responses and 16-byte descriptors
 * ---------------------------------------------------------------------------------------------- */

inline uint8_t sat_u8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

/* common_includes/elas/filter.cpp:416-424 (sobel3x3) = :380-413 (column pass, int16) followed by
 * :235-275 ((1,0,-1) row pass -> "du") and :183-229 ((1,2,1) row pass -> "dv"); both row passes do an
 * arithmetic >>2, add 128 and saturate to u8 (_mm_packus_epi16).  Only x in [1,W-2], y in [1,H-2] can
 * reach a descriptor, so only those are produced; everything else stays 0. */
void sobel_du_dv(const uint8_t *I, int W, int H, int stride, std::vector<uint8_t> &du, std::vector<uint8_t> &dv) {
    du.assign((size_t)W * H, 0);
    dv.assign((size_t)W * H, 0);
    std::vector<int> S((size_t)W), T((size_t)W);
    for (int y = 1; y < H - 1; y++) {
        const uint8_t *r0 = I + (size_t)(y - 1) * stride, *r1 = I + (size_t)y * stride, *r2 = I + (size_t)(y + 1) * stride;
        for (int x = 0; x < W; x++) {
            S[x] = r0[x] + 2 * r1[x] + r2[x]; /* vertical (1,2,1) */
            T[x] = r0[x] - r2[x];             /* vertical (1,0,-1) */
        }
        for (int x = 1; x < W - 1; x++) {
            du[(size_t)y * W + x] = sat_u8(((S[x - 1] - S[x + 1]) >> 2) + 128);
            dv[(size_t)y * W + x] = sat_u8(((T[x - 1] + 2 * T[x] + T[x + 1]) >> 2) + 128);
        }
    }
}

/* common_includes/elas/descriptor.cpp:96-124 (full-resolution branch).  Written for v in [3,H-3),
 * u in [3,W-3); the rest of the (uninitialised in the reference, :31) buffer is canonical zero. */
void descriptor(const uint8_t *I, int W, int H, int stride, uint8_t *desc, bool half_resolution = false) {
    std::vector<uint8_t> du, dv;
    sobel_du_dv(I, W, H, stride, du, dv);
    memset(desc, 0, (size_t)W * H * 16);
    /* descriptor.cpp:48-93: with subsampling only every second line, starting at 4, is computed (same 16 taps) */
    for (int v = half_resolution ? 4 : 3; v < H - 3; v += half_resolution ? 2 : 1) {
        const uint8_t *u0 = &du[(size_t)(v - 2) * W], *u1 = &du[(size_t)(v - 1) * W], *u2 = &du[(size_t)v * W];
        const uint8_t *u3 = &du[(size_t)(v + 1) * W], *u4 = &du[(size_t)(v + 2) * W];
        const uint8_t *w1 = &dv[(size_t)(v - 1) * W], *w2 = &dv[(size_t)v * W], *w3 = &dv[(size_t)(v + 1) * W];
        for (int u = 3; u < W - 3; u++) {
            uint8_t *o = desc + ((size_t)v * W + u) * 16;
            o[0] = u0[u];
            o[1] = u1[u - 2];
            o[2] = u1[u];
            o[3] = u1[u + 2];
            o[4] = u2[u - 1];
            o[5] = u2[u];
            o[6] = u2[u];
            o[7] = u2[u + 1];
            o[8] = u3[u - 2];
            o[9] = u3[u];
            o[10] = u3[u + 2];
            o[11] = u4[u];
            o[12] = w1[u];
            o[13] = w2[u - 1];
            o[14] = w2[u + 1];
            o[15] = w3[u];
        }
    }
}

inline int sad16(const uint8_t *a, const uint8_t *b) {
    int s = 0;
    for (int i = 0; i < 16; i++) s += abs((int)a[i] - (int)b[i]);
    return s;
}

inline int texture16(const uint8_t *a) {
    int s = 0;
    for (int i = 0; i < 16; i++) s += abs((int)a[i] - 128);
    return s;
}

/* ------------------------------------------------------------------------------------------------
 * Stage 2: support points
 * ---------------------------------------------------------------------------------------------- */

/* serial_includes/elas/elas.cpp:266-371 (computeMatchingDisparity). */
int matching_disparity(const elas_params &P, int W, int H, int u, int v, const uint8_t *I1_desc, const uint8_t *I2_desc, bool right_image) {
    const int u_step = 2, v_step = 2, window_size = 3;
    if (!(u >= window_size + u_step && u <= W - window_size - 1 - u_step && v >= window_size + v_step && v <= H - window_size - 1 - v_step))
        return -1; /* :279 */
    const uint8_t *A = right_image ? I2_desc : I1_desc; /* image the block is taken from   (:283-289) */
    const uint8_t *B = right_image ? I1_desc : I2_desc; /* image that is searched */
    if (texture16(A + ((size_t)v * W + u) * 16) < P.support_texture)
        return -1; /* :296-300 */
    const int du[4] = {-u_step, +u_step, -u_step, +u_step}, dv[4] = {-v_step, -v_step, +v_step, +v_step}; /* :271-274 */
    int disp_min_valid = std::max(P.disp_min, 0);
    int disp_max_valid = right_image ? std::min(P.disp_max, W - u - window_size - u_step) : std::min(P.disp_max, u - window_size - u_step); /* :318-323 */
    if (disp_max_valid - disp_min_valid < 10)
        return -1; /* :326 */
    int min_1_E = 32767, min_1_d = -1, min_2_E = 32767, min_2_d = -1;
    for (int d = disp_min_valid; d <= disp_max_valid; d++) {
        int u_warp = right_image ? u + d : u - d;
        int sum = 0;
        for (int c = 0; c < 4; c++) /* :341-349 */
            sum += sad16(A + ((size_t)(v + dv[c]) * W + (u + du[c])) * 16, B + ((size_t)(v + dv[c]) * W + (u_warp + du[c])) * 16);
        if (sum < min_1_E) { /* :352-360 */
            min_2_E = min_1_E;
            min_2_d = min_1_d;
            min_1_E = sum;
            min_1_d = d;
        } else if (sum < min_2_E) {
            min_2_E = sum;
            min_2_d = d;
        }
    }
    if (min_1_d >= 0 && min_2_d >= 0 && (float)min_1_E < P.support_threshold * (float)min_2_E) /* :364 */
        return min_1_d;
    return -1;
}

inline int candidate_step(const elas_params &P) { /* elas.cpp:376-378: an even step at half resolution */
    int step = P.candidate_stepsize;
    if (P.subsampling)
        step += step % 2;
    return step;
}

void candidate_dims(const elas_params &P, int W, int H, int &Wc, int &Hc) { /* elas.cpp:376-386 */
    int step = candidate_step(P);
    Wc = 0;
    Hc = 0;
    for (int u = 0; u < W; u += step) Wc++;
    for (int v = 0; v < H; v += step) Hc++;
}

/* elas.cpp:387-411: calloc'd lattice (row/col 0 stay 0), forward match then backward check. */
void support_raw(const elas_params &P, const uint8_t *d1, const uint8_t *d2, int W, int H, int16_t *dcan) {
    int Wc, Hc;
    candidate_dims(P, W, H, Wc, Hc);
    const int step = candidate_step(P);
    memset(dcan, 0, sizeof(int16_t) * (size_t)Wc * Hc);
    for (int uc = 1; uc < Wc; uc++)
        for (int vc = 1; vc < Hc; vc++) {
            int u = uc * step, v = vc * step;
            int16_t out = -1;
            int d = matching_disparity(P, W, H, u, v, d1, d2, false);
            if (d >= 0) {
                int d2v = matching_disparity(P, W, H, u - d, v, d1, d2, true);
                if (d2v >= 0 && abs(d - d2v) <= P.lr_threshold)
                    out = (int16_t)d;
            }
            dcan[(size_t)vc * Wc + uc] = out;
        }
}

/* elas.cpp:152-176, in place, u outer / v inner. */
void remove_inconsistent(const elas_params &P, int16_t *D, int Wc, int Hc) {
    const int win = P.incon_window_size;
    for (int uc = 0; uc < Wc; uc++)
        for (int vc = 0; vc < Hc; vc++) {
            int d = D[vc * Wc + uc];
            if (d < 0)
                continue;
            int support = 0;
            for (int u2 = uc - win; u2 <= uc + win; u2++)
                for (int v2 = vc - win; v2 <= vc + win; v2++)
                    if (u2 >= 0 && v2 >= 0 && u2 < Wc && v2 < Hc) {
                        int d2 = D[v2 * Wc + u2];
                        if (d2 >= 0 && abs(d - d2) <= P.incon_threshold)
                            support++;
                    }
            if (support < P.incon_min_support)
                D[vc * Wc + uc] = -1;
        }
}

/* elas.cpp:178-233, in place. */
void remove_redundant(int16_t *D, int Wc, int Hc, int max_dist, int thr, bool vertical) {
    const int dir_u[2] = {vertical ? 0 : -1, vertical ? 0 : +1};
    const int dir_v[2] = {vertical ? -1 : 0, vertical ? +1 : 0};
    for (int uc = 0; uc < Wc; uc++)
        for (int vc = 0; vc < Hc; vc++) {
            int d = D[vc * Wc + uc];
            if (d < 0)
                continue;
            bool redundant = true;
            for (int i = 0; i < 2 && redundant; i++) {
                int u2 = uc, v2 = vc;
                bool support = false;
                for (int j = 0; j < max_dist; j++) {
                    u2 += dir_u[i];
                    v2 += dir_v[i];
                    if (u2 < 0 || v2 < 0 || u2 >= Wc || v2 >= Hc)
                        break;
                    int d2 = D[v2 * Wc + u2];
                    if (d2 >= 0 && abs(d - d2) <= thr) {
                        support = true;
                        break;
                    }
                }
                if (!support)
                    redundant = false;
            }
            if (redundant)
                D[vc * Wc + uc] = -1;
        }
}

struct Pt {
    int32_t u, v, d;
};

/* elas.cpp:413-433 + addCornerSupportPoints :235-264. */
std::vector<Pt> support_filter(const elas_params &P, int16_t *dcan, int W, int H) {
    int Wc, Hc;
    candidate_dims(P, W, H, Wc, Hc);
    remove_inconsistent(P, dcan, Wc, Hc);
    remove_redundant(dcan, Wc, Hc, 5, 1, true);
    remove_redundant(dcan, Wc, Hc, 5, 1, false);
    std::vector<Pt> s;
    const int step = candidate_step(P);
    for (int uc = 1; uc < Wc; uc++)
        for (int vc = 1; vc < Hc; vc++)
            if (dcan[vc * Wc + uc] >= 0)
                s.push_back(Pt{uc * step, vc * step, dcan[vc * Wc + uc]});
    if (P.add_corners) {
        Pt b[4] = {{0, 0, 0}, {0, H - 1, 0}, {W - 1, 0, 0}, {W - 1, H - 1, 0}};
        for (int i = 0; i < 4; i++) {
            int best = 10000000;
            for (size_t j = 0; j < s.size(); j++) {
                int du = b[i].u - s[j].u, dv = b[i].v - s[j].v;
                int dist = du * du + dv * dv;
                if (dist < best) {
                    best = dist;
                    b[i].d = s[j].d;
                }
            }
        }
        Pt r2 = {b[2].u + b[2].d, b[2].v, b[2].d}, r3 = {b[3].u + b[3].d, b[3].v, b[3].d};
        for (int i = 0; i < 4; i++) s.push_back(b[i]);
        s.push_back(r2);
        s.push_back(r3);
    }
    return s;
}

/* ------------------------------------------------------------------------------------------------
 * Stage 3: Delaunay triangulation
 *
 * The reference calls Shewchuk's Triangle 1.6 with switches "zQB" (elas.cpp:483-484): divide and
 * conquer with alternating (Dwyer) cuts and exact predicates.  Support points sit on an integer
 * lattice, so co-circular quadruples are the norm and the triangulation is not unique: to get the same
 * triangles, in the same order, with the same corner order, the restatement follows the same sequence
 * of topological operations:
 *   common_includes/elas/triangle.cpp:5183-5229 (vertexsort)      :5243-5294 (vertexmedian)
 *   :5307-5325 (alternateaxes)  :5362-5651 (mergehulls)  :5670-5815 (divconqrecurse)
 *   :5817-5859 (removeghosts)   :5871-5924 (divconqdelaunay)  :3833-3836 (randomnation)
 *   :7449-7500 (writeelements: live triangles in pool order, corners org/dest/apex at orientation 0)
 * Differences in representation: triangles are slots of two flat int arrays (3 neighbour codes, 3 vertex
 * ids) instead of pointer blocks; vertex ids are input indices; "NULL" is -1; and because every
 * coordinate is an integer, the adaptive-precision predicates (:2487-3200) are evaluated exactly in
 * 64-bit integers, which yields the same signs.
 * ---------------------------------------------------------------------------------------------- */
namespace dc {

const int NEXT3[3] = {1, 2, 0}, PREV3[3] = {2, 0, 1};

struct Edge { /* "oriented triangle": triangle slot + which of its three edges */
    int t, o;
};

struct Mesh {
    const int64_t *X, *Y;
    std::vector<int> nbr, vtx; /* 3 per slot */
    std::vector<char> dead;
    unsigned long seed = 1; /* triangle.cpp:3818 */
    int n_slots = 0;

    Mesh(const int64_t *x, const int64_t *y, int n) : X(x), Y(y) {
        nbr.reserve(12 * (size_t)n);
        vtx.reserve(12 * (size_t)n);
        make(); /* slot 0 plays the role of "outer space" (dummytri) */
    }
    Edge make() { /* maketriangle(), :2068-2101 */
        int t = n_slots++;
        for (int i = 0; i < 3; i++) {
            nbr.push_back(0);
            vtx.push_back(-1);
        }
        dead.push_back(0);
        return Edge{t, 0};
    }
    Edge sym(Edge e) const {
        int c = nbr[3 * e.t + e.o];
        return Edge{c >> 2, c & 3};
    }
    static Edge lnext(Edge e) { return Edge{e.t, NEXT3[e.o]}; }
    static Edge lprev(Edge e) { return Edge{e.t, PREV3[e.o]}; }
    int org(Edge e) const { return vtx[3 * e.t + NEXT3[e.o]]; }
    int dest(Edge e) const { return vtx[3 * e.t + PREV3[e.o]]; }
    int apex(Edge e) const { return vtx[3 * e.t + e.o]; }
    void setorg(Edge e, int v) { vtx[3 * e.t + NEXT3[e.o]] = v; }
    void setdest(Edge e, int v) { vtx[3 * e.t + PREV3[e.o]] = v; }
    void setapex(Edge e, int v) { vtx[3 * e.t + e.o] = v; }
    void bond(Edge a, Edge b) {
        nbr[3 * a.t + a.o] = (b.t << 2) | b.o;
        nbr[3 * b.t + b.o] = (a.t << 2) | a.o;
    }
    /* exact orientation / in-circle signs (stand in for :2563-2603 and :3133-3190) */
    int64_t ccw(int a, int b, int c) const { return (X[a] - X[c]) * (Y[b] - Y[c]) - (Y[a] - Y[c]) * (X[b] - X[c]); }
    int64_t incircle(int a, int b, int c, int d) const {
        int64_t adx = X[a] - X[d], ady = Y[a] - Y[d], bdx = X[b] - X[d], bdy = Y[b] - Y[d], cdx = X[c] - X[d], cdy = Y[c] - Y[d];
        int64_t al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
        return al * (bdx * cdy - cdx * bdy) + bl * (cdx * ady - adx * cdy) + cl * (adx * bdy - bdx * ady);
    }
    unsigned long randomnation(unsigned int choices) { /* :3833-3836 */
        seed = (seed * 1366ul + 150889ul) % 714025ul;
        return seed / (714025ul / choices + 1);
    }
    bool less_xy(int a, int64_t px, int64_t py) const { return X[a] < px || (X[a] == px && Y[a] < py); }
    bool greater_xy(int a, int64_t px, int64_t py) const { return X[a] > px || (X[a] == px && Y[a] > py); }

    void vertexsort(int *a, int n) { /* :5183-5229 */
        if (n == 2) {
            if (greater_xy(a[0], X[a[1]], Y[a[1]]))
                std::swap(a[0], a[1]);
            return;
        }
        int pivot = (int)randomnation((unsigned)n);
        int64_t px = X[a[pivot]], py = Y[a[pivot]];
        int left = -1, right = n;
        while (left < right) {
            do {
                left++;
            } while (left <= right && less_xy(a[left], px, py));
            do {
                right--;
            } while (left <= right && greater_xy(a[right], px, py));
            if (left < right)
                std::swap(a[left], a[right]);
        }
        if (left > 1)
            vertexsort(a, left);
        if (right < n - 2)
            vertexsort(a + right + 1, n - right - 1);
    }
    int64_t key1(int v, int axis) const { return axis == 0 ? X[v] : Y[v]; }
    int64_t key2(int v, int axis) const { return axis == 0 ? Y[v] : X[v]; }
    void vertexmedian(int *a, int n, int median, int axis) { /* :5243-5294 */
        if (n == 2) {
            if (key1(a[0], axis) > key1(a[1], axis) || (key1(a[0], axis) == key1(a[1], axis) && key2(a[0], axis) > key2(a[1], axis)))
                std::swap(a[0], a[1]);
            return;
        }
        int pivot = (int)randomnation((unsigned)n);
        int64_t p1 = key1(a[pivot], axis), p2 = key2(a[pivot], axis);
        int left = -1, right = n;
        while (left < right) {
            do {
                left++;
            } while (left <= right && (key1(a[left], axis) < p1 || (key1(a[left], axis) == p1 && key2(a[left], axis) < p2)));
            do {
                right--;
            } while (left <= right && (key1(a[right], axis) > p1 || (key1(a[right], axis) == p1 && key2(a[right], axis) > p2)));
            if (left < right)
                std::swap(a[left], a[right]);
        }
        if (left > median)
            vertexmedian(a, left, median, axis);
        if (right < median - 1)
            vertexmedian(a + right + 1, n - right - 1, median - right - 1, axis);
    }
    void alternateaxes(int *a, int n, int axis) { /* :5307-5325 */
        int divider = n >> 1;
        if (n <= 3)
            axis = 0;
        vertexmedian(a, n, divider, axis);
        if (n - divider >= 2) {
            if (divider >= 2)
                alternateaxes(a, divider, 1 - axis);
            alternateaxes(a + divider, n - divider, 1 - axis);
        }
    }

    /* :5362-5651.  Knits the left and right hulls together from the lower common tangent upwards. */
    void mergehulls(Edge &farleft, Edge &innerleft, Edge &innerright, Edge &farright, int axis) {
        int innerleftdest = dest(innerleft), innerleftapex = apex(innerleft);
        int innerrightorg = org(innerright), innerrightapex = apex(innerright);
        if (axis == 1) { /* horizontal cut: move the extreme handles to the top-/bottom-most vertices (:5393-5431) */
            int farleftpt = org(farleft), farleftapex = apex(farleft);
            int farrightpt = dest(farright), farrightapex = apex(farright);
            while (Y[farleftapex] < Y[farleftpt]) {
                farleft = sym(lnext(farleft));
                farleftpt = farleftapex;
                farleftapex = apex(farleft);
            }
            Edge check = sym(innerleft);
            int checkvertex = apex(check);
            while (Y[checkvertex] > Y[innerleftdest]) {
                innerleft = lnext(check);
                innerleftapex = innerleftdest;
                innerleftdest = checkvertex;
                check = sym(innerleft);
                checkvertex = apex(check);
            }
            while (Y[innerrightapex] < Y[innerrightorg]) {
                innerright = sym(lnext(innerright));
                innerrightorg = innerrightapex;
                innerrightapex = apex(innerright);
            }
            check = sym(farright);
            checkvertex = apex(check);
            while (Y[checkvertex] > Y[farrightpt]) {
                farright = lnext(check);
                farrightapex = farrightpt;
                farrightpt = checkvertex;
                check = sym(farright);
                checkvertex = apex(check);
            }
            (void)farrightapex;
        }
        /* lower common tangent (:5433-5451) */
        bool changed;
        do {
            changed = false;
            if (ccw(innerleftdest, innerleftapex, innerrightorg) > 0) {
                innerleft = sym(lprev(innerleft));
                innerleftdest = innerleftapex;
                innerleftapex = apex(innerleft);
                changed = true;
            }
            if (ccw(innerrightapex, innerrightorg, innerleftdest) > 0) {
                innerright = sym(lnext(innerright));
                innerrightorg = innerrightapex;
                innerrightapex = apex(innerright);
                changed = true;
            }
        } while (changed);
        Edge leftcand = sym(innerleft), rightcand = sym(innerright);
        /* bottom bounding triangle (:5456-5463) */
        Edge base = make();
        bond(base, innerleft);
        base = lnext(base);
        bond(base, innerright);
        base = lnext(base);
        setorg(base, innerrightorg);
        setdest(base, innerleftdest);
        if (innerleftdest == org(farleft)) /* :5470-5477 */
            farleft = lnext(base);
        if (innerrightorg == dest(farright))
            farright = lprev(base);
        int lowerleft = innerleftdest, lowerright = innerrightorg;
        int upperleft = apex(leftcand), upperright = apex(rightcand);
        for (;;) {
            bool leftfinished = ccw(upperleft, lowerleft, lowerright) <= 0;
            bool rightfinished = ccw(upperright, lowerleft, lowerright) <= 0;
            if (leftfinished && rightfinished) { /* top bounding triangle (:5492-5533) */
                Edge top = make();
                setorg(top, lowerleft);
                setdest(top, lowerright);
                bond(top, base);
                top = lnext(top);
                bond(top, rightcand);
                top = lnext(top);
                bond(top, leftcand);
                if (axis == 1) { /* restore left-/right-most handles (:5509-5532) */
                    int farleftpt = org(farleft);
                    int farrightpt = dest(farright), farrightapex = apex(farright);
                    Edge check = sym(farleft);
                    int checkvertex = apex(check);
                    while (X[checkvertex] < X[farleftpt]) {
                        farleft = lprev(check);
                        farleftpt = checkvertex;
                        check = sym(farleft);
                        checkvertex = apex(check);
                    }
                    while (X[farrightapex] > X[farrightpt]) {
                        farright = sym(lprev(farright));
                        farrightpt = farrightapex;
                        farrightapex = apex(farright);
                    }
                }
                return;
            }
            if (!leftfinished) { /* eat non-Delaunay edges of the left triangulation (:5536-5580) */
                Edge next = sym(lprev(leftcand));
                int nextapex = apex(next);
                if (nextapex != -1) {
                    bool bad = incircle(lowerleft, lowerright, upperleft, nextapex) > 0;
                    while (bad) {
                        next = lnext(next);
                        Edge topcasing = sym(next);
                        next = lnext(next);
                        Edge sidecasing = sym(next);
                        bond(next, topcasing);
                        bond(leftcand, sidecasing);
                        leftcand = lnext(leftcand);
                        Edge outercasing = sym(leftcand);
                        next = lprev(next);
                        bond(next, outercasing);
                        setorg(leftcand, lowerleft);
                        setdest(leftcand, -1);
                        setapex(leftcand, nextapex);
                        setorg(next, -1);
                        setdest(next, upperleft);
                        setapex(next, nextapex);
                        upperleft = nextapex;
                        next = sidecasing;
                        nextapex = apex(next);
                        bad = (nextapex != -1) && incircle(lowerleft, lowerright, upperleft, nextapex) > 0;
                    }
                }
            }
            if (!rightfinished) { /* same for the right triangulation (:5582-5626) */
                Edge next = sym(lnext(rightcand));
                int nextapex = apex(next);
                if (nextapex != -1) {
                    bool bad = incircle(lowerleft, lowerright, upperright, nextapex) > 0;
                    while (bad) {
                        next = lprev(next);
                        Edge topcasing = sym(next);
                        next = lprev(next);
                        Edge sidecasing = sym(next);
                        bond(next, topcasing);
                        bond(rightcand, sidecasing);
                        rightcand = lprev(rightcand);
                        Edge outercasing = sym(rightcand);
                        next = lnext(next);
                        bond(next, outercasing);
                        setorg(rightcand, -1);
                        setdest(rightcand, lowerright);
                        setapex(rightcand, nextapex);
                        setorg(next, upperright);
                        setdest(next, -1);
                        setapex(next, nextapex);
                        upperright = nextapex;
                        next = sidecasing;
                        nextapex = apex(next);
                        bad = (nextapex != -1) && incircle(lowerleft, lowerright, upperright, nextapex) > 0;
                    }
                }
            }
            if (leftfinished || (!rightfinished && incircle(upperleft, lowerleft, lowerright, upperright) > 0)) { /* :5627-5645 */
                bond(base, rightcand);
                base = lprev(rightcand);
                setdest(base, lowerleft);
                lowerright = upperright;
                rightcand = sym(base);
                upperright = apex(rightcand);
            } else {
                bond(base, leftcand);
                base = lnext(leftcand);
                setorg(base, lowerright);
                lowerleft = upperleft;
                leftcand = sym(base);
                upperleft = apex(leftcand);
            }
        }
    }

    void recurse(int *a, int n, int axis, Edge &farleft, Edge &farright) { /* :5670-5815 */
        if (n == 2) {
            farleft = make();
            setorg(farleft, a[0]);
            setdest(farleft, a[1]);
            farright = make();
            setorg(farright, a[1]);
            setdest(farright, a[0]);
            bond(farleft, farright);
            farleft = lprev(farleft);
            farright = lnext(farright);
            bond(farleft, farright);
            farleft = lprev(farleft);
            farright = lnext(farright);
            bond(farleft, farright);
            farleft = lprev(farright);
        } else if (n == 3) {
            Edge mid = make(), t1 = make(), t2 = make(), t3 = make();
            int64_t area = ccw(a[0], a[1], a[2]);
            if (area == 0) { /* collinear: two edges, four bounding triangles (:5715-5743) */
                setorg(mid, a[0]);
                setdest(mid, a[1]);
                setorg(t1, a[1]);
                setdest(t1, a[0]);
                setorg(t2, a[2]);
                setdest(t2, a[1]);
                setorg(t3, a[1]);
                setdest(t3, a[2]);
                bond(mid, t1);
                bond(t2, t3);
                mid = lnext(mid);
                t1 = lprev(t1);
                t2 = lnext(t2);
                t3 = lprev(t3);
                bond(mid, t3);
                bond(t1, t2);
                mid = lnext(mid);
                t1 = lprev(t1);
                t2 = lnext(t2);
                t3 = lprev(t3);
                bond(mid, t1);
                bond(t2, t3);
                farleft = t1;
                farright = t2;
            } else { /* one real triangle `mid` + three bounding triangles (:5744-5791) */
                setorg(mid, a[0]);
                setdest(t1, a[0]);
                setorg(t3, a[0]);
                int second = area > 0 ? a[1] : a[2], third = area > 0 ? a[2] : a[1];
                setdest(mid, second);
                setorg(t1, second);
                setdest(t2, second);
                setapex(mid, third);
                setorg(t2, third);
                setdest(t3, third);
                bond(mid, t1);
                mid = lnext(mid);
                bond(mid, t2);
                mid = lnext(mid);
                bond(mid, t3);
                t1 = lprev(t1);
                t2 = lnext(t2);
                bond(t1, t2);
                t1 = lprev(t1);
                t3 = lprev(t3);
                bond(t1, t3);
                t2 = lnext(t2);
                t3 = lprev(t3);
                bond(t2, t3);
                farleft = t1;
                farright = area > 0 ? t2 : lnext(farleft);
            }
        } else {
            int divider = n >> 1;
            Edge innerleft, innerright;
            recurse(a, divider, 1 - axis, farleft, innerleft);
            recurse(a + divider, n - divider, 1 - axis, innerright, farright);
            mergehulls(farleft, innerleft, innerright, farright, axis);
        }
    }

    void removeghosts(Edge start) { /* :5817-5859 */
        Edge dissolve = start;
        do {
            Edge deadtri = lnext(dissolve);
            dissolve = sym(lprev(dissolve));
            nbr[3 * dissolve.t + dissolve.o] = 0; /* dissolve(): now borders outer space */
            dissolve = sym(deadtri);
            dead[deadtri.t] = 1;
        } while (!(dissolve.t == start.t && dissolve.o == start.o));
    }
};

/* :5871-5924 + :7449-7500.  Points must hold integers (they do: elas.cpp:452-460 stores int32 into float). */
int triangulate(const float *xy, int n, std::vector<int32_t> &out) {
    out.clear();
    if (n < 3)
        return 0;
    std::vector<int64_t> X(n), Y(n);
    for (int i = 0; i < n; i++) {
        X[i] = (int64_t)xy[2 * i];
        Y[i] = (int64_t)xy[2 * i + 1];
    }
    Mesh m(X.data(), Y.data(), n);
    std::vector<int> a(n);
    for (int i = 0; i < n; i++) a[i] = i;
    m.vertexsort(a.data(), n);
    int i = 0;
    for (int j = 1; j < n; j++) { /* drop duplicates (:5890-5903) */
        if (X[a[i]] == X[a[j]] && Y[a[i]] == Y[a[j]])
            continue;
        a[++i] = a[j];
    }
    i++;
    int divider = i >> 1;
    if (i - divider >= 2) { /* :5904-5913 */
        if (divider >= 2)
            m.alternateaxes(a.data(), divider, 1);
        m.alternateaxes(a.data() + divider, i - divider, 1);
    }
    if (i < 2)
        return 0;
    Edge hullleft, hullright;
    m.recurse(a.data(), i, 0, hullleft, hullright);
    m.removeghosts(hullleft);
    for (int t = 1; t < m.n_slots; t++) {
        if (m.dead[t])
            continue;
        out.push_back(m.vtx[3 * t + 1]); /* org  at orientation 0 */
        out.push_back(m.vtx[3 * t + 2]); /* dest */
        out.push_back(m.vtx[3 * t + 0]); /* apex */
    }
    return (int)out.size() / 3;
}

}  // namespace dc

/* elas.cpp:442-501 */
std::vector<int32_t> delaunay(const std::vector<Pt> &s, bool right_image) {
    std::vector<float> xy(2 * s.size());
    for (size_t i = 0; i < s.size(); i++) {
        xy[2 * i] = (float)(right_image ? s[i].u - s[i].d : s[i].u);
        xy[2 * i + 1] = (float)s[i].v;
    }
    std::vector<int32_t> tri;
    dc::triangulate(xy.data(), (int)s.size(), tri);
    return tri;
}

/* ------------------------------------------------------------------------------------------------
 * Stage 4: disparity planes
 * ---------------------------------------------------------------------------------------------- */

/* common_includes/elas/matrix.cpp:418-510 (Gauss-Jordan, full pivoting, eps = 1e-20 from matrix.h:112),
 * specialised to a 3x3 system with one right-hand side, in double as the reference (matrix.h:46). */
bool solve3(double A[3][3], double B[3]) {
    int ipiv[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) {
        double big = 0.0;
        int irow = 0, icol = 0;
        for (int j = 0; j < 3; j++)
            if (ipiv[j] != 1)
                for (int k = 0; k < 3; k++)
                    if (ipiv[k] == 0)
                        if (fabs(A[j][k]) >= big) {
                            big = fabs(A[j][k]);
                            irow = j;
                            icol = k;
                        }
        ++ipiv[icol];
        if (irow != icol) {
            for (int l = 0; l < 3; l++) std::swap(A[irow][l], A[icol][l]);
            std::swap(B[irow], B[icol]);
        }
        if (fabs(A[icol][icol]) < 1e-20)
            return false;
        double pivinv = 1.0 / A[icol][icol];
        A[icol][icol] = 1.0;
        for (int l = 0; l < 3; l++) A[icol][l] *= pivinv;
        B[icol] *= pivinv;
        for (int ll = 0; ll < 3; ll++)
            if (ll != icol) {
                double dum = A[ll][icol];
                A[ll][icol] = 0.0;
                for (int l = 0; l < 3; l++) A[ll][l] -= A[icol][l] * dum;
                B[ll] -= B[icol] * dum;
            }
    }
    return true;
}

/* elas.cpp:503-575 */
void planes(const int32_t *s, const int32_t *tri, int nt, float *out) {
    for (int i = 0; i < nt; i++) {
        const int32_t *c[3] = {s + 3 * tri[3 * i], s + 3 * tri[3 * i + 1], s + 3 * tri[3 * i + 2]};
        for (int side = 0; side < 2; side++) {
            double A[3][3], B[3];
            for (int r = 0; r < 3; r++) {
                A[r][0] = side == 0 ? c[r][0] : c[r][0] - c[r][2];
                A[r][1] = c[r][1];
                A[r][2] = 1;
                B[r] = c[r][2];
            }
            float *o = out + 6 * i + 3 * side;
            if (solve3(A, B)) {
                o[0] = (float)B[0];
                o[1] = (float)B[1];
                o[2] = (float)B[2];
            } else {
                o[0] = o[1] = o[2] = 0;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Stage 5: candidate grid
 * ---------------------------------------------------------------------------------------------- */

void grid_dims_of(const elas_params &P, int W, int H, int32_t dims[3]) { /* elas.cpp:88-90 */
    dims[0] = P.disp_max + 2;
    dims[1] = (int32_t)ceil((float)W / (float)P.grid_size);
    dims[2] = (int32_t)ceil((float)H / (float)P.grid_size);
}

/* elas.cpp:577-653.  The 3x3 dilation runs over the *flat* cell array (:613-628), which leaves the first and
 * last cell rows (and two more cells) empty and wraps at row ends; reproduced literally. */
void grid(const elas_params &P, const int32_t *s, int n, int W, int H, bool right_image, int32_t *out) {
    int32_t gd[3];
    grid_dims_of(P, W, H, gd);
    const int gw = gd[1], gh = gd[2], D = P.disp_max + 1;
    std::vector<int32_t> t1((size_t)D * gw * gh, 0), t2((size_t)D * gw * gh, 0);
    for (int i = 0; i < n; i++) {
        int x_curr = s[3 * i], y_curr = s[3 * i + 1], d_curr = s[3 * i + 2];
        int d_min = std::max(d_curr - 1, 0), d_max = std::min(d_curr + 1, P.disp_max);
        for (int d = d_min; d <= d_max; d++) {
            int x;
            if (!right_image)
                x = (int)floor((float)(x_curr / P.grid_size));
            else
                x = (int)floor((float)(x_curr - d_curr) / (float)P.grid_size);
            int y = (int)floor((float)y_curr / (float)P.grid_size);
            if (x >= 0 && x < gw && y >= 0 && y < gh)
                t1[((size_t)y * gw + x) * D + d] = 1;
        }
    }
    const long total = (long)gw * gh * D;
    for (long r = (long)(gw + 1) * D, k = 0; (2L * gw + 2) * D + k < total; r++, k++) {
        long b = k; /* index of the top-left operand */
        t2[r] = t1[b] | t1[b + D] | t1[b + 2 * D] | t1[b + (long)gw * D] | t1[b + (long)(gw + 1) * D] | t1[b + (long)(gw + 2) * D] |
                t1[b + 2L * gw * D] | t1[b + (2L * gw + 1) * D] | t1[b + (2L * gw + 2) * D];
    }
    memset(out, 0, sizeof(int32_t) * (size_t)(P.disp_max + 2) * gw * gh);
    for (int x = 0; x < gw; x++)
        for (int y = 0; y < gh; y++) {
            int32_t *cell = out + ((size_t)y * gw + x) * (P.disp_max + 2);
            int cnt = 0;
            for (int d = 0; d <= P.disp_max; d++)
                if (t2[((size_t)y * gw + x) * D + d] > 0)
                    cell[++cnt] = d;
            cell[0] = cnt;
        }
}

/* ------------------------------------------------------------------------------------------------
 * Stage 6: dense matching
 * ---------------------------------------------------------------------------------------------- */

/* x86-64 evaluates `(uint32_t)float_expr` (elas.cpp:916-917,931-932) as a 64-bit truncation whose low
 * 32 bits are kept; stored into an int32 this is plain truncation towards zero for every value that can occur. */
inline int32_t f2u2i(float f) { return (int32_t)(int64_t)f; }

struct DenseCtx {
    const elas_params &P;
    int W, H;
    const int32_t *grid;
    int32_t gd[3];
    const uint8_t *I1_desc, *I2_desc;
    std::vector<int32_t> prior;
    int plane_radius;
    bool right_image;
    float *D;
};

/* elas.cpp:688-801 (findMatch) with :655-686 (updatePosteriorMinimum) folded in. */
void find_match(DenseCtx &c, int u, int v, float plane_a, float plane_b, float plane_c, bool valid) {
    const int W = c.W, H = c.H;
    const int disp_num = c.gd[0] - 1;
    const int window_size = 2;
    if (u < window_size || u >= W - window_size)
        return;
    size_t line = (size_t)W * std::max(std::min(v, H - 3), 2);
    const uint8_t *A = (c.right_image ? c.I2_desc : c.I1_desc) + line * 16;
    const uint8_t *B = (c.right_image ? c.I1_desc : c.I2_desc) + line * 16;
    const uint8_t *blk = A + (size_t)16 * u;
    if (texture16(blk) < c.P.match_texture)
        return;
    int d_plane = (int32_t)(plane_a * (float)u + plane_b * (float)v + plane_c);
    int d_plane_min = std::max(d_plane - c.plane_radius, 0);
    int d_plane_max = std::min(d_plane + c.plane_radius, disp_num - 1);
    int grid_x = (int)floor((float)u / (float)c.P.grid_size);
    int grid_y = (int)floor((float)v / (float)c.P.grid_size);
    const int32_t *cell = c.grid + ((size_t)grid_y * c.gd[1] + grid_x) * c.gd[0];
    int num_grid = cell[0];
    int min_val = 10000, min_d = -1;
    for (int i = 0; i < num_grid; i++) {
        int d_curr = cell[1 + i];
        if (d_curr < d_plane_min || d_curr > d_plane_max) {
            int u_warp = c.right_image ? u + d_curr : u - d_curr;
            if (u_warp < window_size || u_warp >= W - window_size)
                continue;
            int val = sad16(blk, B + (size_t)16 * u_warp);
            if (val < min_val) {
                min_val = val;
                min_d = d_curr;
            }
        }
    }
    for (int d_curr = d_plane_min; d_curr <= d_plane_max; d_curr++) {
        int u_warp = c.right_image ? u + d_curr : u - d_curr;
        if (u_warp < window_size || u_warp >= W - window_size)
            continue;
        int val = sad16(blk, B + (size_t)16 * u_warp) + (valid ? c.prior[abs(d_curr - d_plane)] : 0);
        if (val < min_val) {
            min_val = val;
            min_d = d_curr;
        }
    }
    /* :707-711: at half resolution the result lands at (u/2, v/2) of a (W/2)-wide map */
    const size_t d_addr = c.P.subsampling ? (size_t)(v / 2) * (W / 2) + u / 2 : (size_t)v * W + u;
    c.D[d_addr] = min_d >= 0 ? (float)min_d : -1.0f;
}

/* elas.cpp:804-944 (computeDisparity). */
void dense(const elas_params &P, const int32_t *s, const int32_t *tri, const float *pl, int nt, const int32_t *grid_, const uint8_t *d1,
           const uint8_t *d2, int W, int H, bool right_image, float *D) {
    DenseCtx c{P, W, H, grid_, {0, 0, 0}, d1, d2, {}, 0, right_image, D};
    grid_dims_of(P, W, H, c.gd);
    const int disp_num = c.gd[0] - 1;
    const bool sub = P.subsampling != 0;
    const size_t NM = sub ? (size_t)(W / 2) * (H / 2) : (size_t)W * H; /* :819-825 */
    for (size_t i = 0; i < NM; i++) D[i] = -10;
    float two_sigma_squared = 2 * P.sigma * P.sigma;
    c.prior.resize(disp_num);
    for (int delta_d = 0; delta_d < disp_num; delta_d++) /* :830-831, float expf/logf as the reference resolves them */
        c.prior[delta_d] = (int32_t)((-logf(P.gamma + expf(-delta_d * delta_d / two_sigma_squared)) + logf(P.gamma)) / P.beta);
    c.plane_radius = (int32_t)std::max((float)ceil(P.sigma * P.sradius), (float)2.0);

    for (int i = 0; i < nt; i++) {
        const float *t = pl + 6 * i;
        float plane_a, plane_b, plane_c, plane_d;
        if (!right_image) {
            plane_a = t[0], plane_b = t[1], plane_c = t[2], plane_d = t[3];
        } else {
            plane_a = t[3], plane_b = t[4], plane_c = t[5], plane_d = t[0];
        }
        float tri_u[3], tri_v[3];
        for (int k = 0; k < 3; k++) {
            const int32_t *pt = s + 3 * tri[3 * i + k];
            tri_u[k] = right_image ? (float)(pt[0] - pt[2]) : (float)pt[0];
            tri_v[k] = (float)pt[1];
        }
        for (int j = 0; j < 3; j++) /* :873-884 */
            for (int k = 0; k < j; k++)
                if (tri_u[k] > tri_u[j]) {
                    std::swap(tri_u[j], tri_u[k]);
                    std::swap(tri_v[j], tri_v[k]);
                }
        float A_u = tri_u[0], A_v = tri_v[0], B_u = tri_u[1], B_v = tri_v[1], C_u = tri_u[2], C_v = tri_v[2];
        float AB_a = 0, AC_a = 0, BC_a = 0;
        if ((int32_t)A_u != (int32_t)B_u)
            AB_a = (A_v - B_v) / (A_u - B_u);
        if ((int32_t)A_u != (int32_t)C_u)
            AC_a = (A_v - C_v) / (A_u - C_u);
        if ((int32_t)B_u != (int32_t)C_u)
            BC_a = (B_v - C_v) / (B_u - C_u);
        float AB_b = A_v - AB_a * A_u, AC_b = A_v - AC_a * A_u, BC_b = B_v - BC_a * B_u;
        bool valid = fabs(plane_a) < 0.7 && fabs(plane_d) < 0.7; /* float |.| promoted to double against 0.7 (:910) */

        if ((int32_t)A_u != (int32_t)B_u)
            for (int u = std::max((int32_t)A_u, 0); u < std::min((int32_t)B_u, W); u++) {
                if (sub && u % 2 != 0) /* :915 */
                    continue;
                int v_1 = f2u2i(AC_a * (float)u + AC_b), v_2 = f2u2i(AB_a * (float)u + AB_b);
                for (int v = std::min(v_1, v_2); v < std::max(v_1, v_2); v++)
                    if (!sub || v % 2 == 0) /* :919 */
                        find_match(c, u, v, plane_a, plane_b, plane_c, valid);
            }
        if ((int32_t)B_u != (int32_t)C_u)
            for (int u = std::max((int32_t)B_u, 0); u < std::min((int32_t)C_u, W); u++) {
                if (sub && u % 2 != 0) /* :930 */
                    continue;
                int v_1 = f2u2i(AC_a * (float)u + AC_b), v_2 = f2u2i(BC_a * (float)u + BC_b);
                for (int v = std::min(v_1, v_2); v < std::max(v_1, v_2); v++)
                    if (!sub || v % 2 == 0) /* :934 */
                        find_match(c, u, v, plane_a, plane_b, plane_c, valid);
            }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Stages 7-11: post-processing
 * ---------------------------------------------------------------------------------------------- */

/* elas.cpp:946-1011 */
void lr_check(const elas_params &P, float *D1, float *D2, int W, int H) { /* W, H: map size (half the image at half resolution) */
    std::vector<float> C1(D1, D1 + (size_t)W * H), C2(D2, D2 + (size_t)W * H);
    for (int u = 0; u < W; u++)
        for (int v = 0; v < H; v++) {
            size_t addr = (size_t)v * W + u;
            float d1 = C1[addr], d2 = C2[addr];
            float u_warp_1, u_warp_2;
            if (P.subsampling) { /* :972-975: disparities stay in full-resolution pixels */
                u_warp_1 = (float)u - d1 / 2;
                u_warp_2 = (float)u + d2 / 2;
            } else {
                u_warp_1 = (float)u - d1;
                u_warp_2 = (float)u + d2;
            }
            if (d1 >= 0 && u_warp_1 >= 0 && u_warp_1 < W) {
                if (fabs(C2[(size_t)v * W + (int32_t)u_warp_1] - d1) > P.lr_threshold)
                    D1[addr] = -10;
            } else
                D1[addr] = -10;
            if (d2 >= 0 && u_warp_2 >= 0 && u_warp_2 < W) {
                if (fabs(C1[(size_t)v * W + (int32_t)u_warp_2] - d2) > P.lr_threshold)
                    D2[addr] = -10;
            } else
                D2[addr] = -10;
        }
}

/* elas.cpp:1013-1124: breadth-first flood fill seeded at every unvisited pixel, u outer / v inner. */
void speckle(const elas_params &P, float *D, int W, int H) {
    const int min_size = P.subsampling ? (int32_t)(sqrt((float)P.speckle_size) * 2) : P.speckle_size; /* :1017-1022 */
    std::vector<int32_t> done((size_t)W * H, 0), lu((size_t)W * H), lv((size_t)W * H);
    for (int u = 0; u < W; u++)
        for (int v = 0; v < H; v++) {
            if (done[(size_t)v * W + u])
                continue;
            lu[0] = u;
            lv[0] = v;
            int count = 1, curr = 0;
            while (curr < count) {
                int uc = lu[curr], vc = lv[curr];
                size_t addr_curr = (size_t)vc * W + uc;
                const int nu[4] = {uc - 1, uc + 1, uc, uc}, nv[4] = {vc, vc, vc - 1, vc + 1};
                for (int i = 0; i < 4; i++)
                    if (nu[i] >= 0 && nv[i] >= 0 && nu[i] < W && nv[i] < H) {
                        size_t an = (size_t)nv[i] * W + nu[i];
                        if (done[an] == 0 && D[an] >= 0 && fabs(D[addr_curr] - D[an]) <= P.speckle_sim_threshold) {
                            lu[count] = nu[i];
                            lv[count] = nv[i];
                            count++;
                            done[an] = 1;
                        }
                    }
                curr++;
                done[addr_curr] = 1;
            }
            if (count < min_size)
                for (int i = 0; i < count; i++) D[(size_t)lv[i] * W + lu[i]] = -10;
        }
}

/* elas.cpp:1126-1294 */
void gap(const elas_params &P, float *D, int W, int H) {
    const int gw = P.subsampling ? P.ipol_gap_width / 2 + 1 : P.ipol_gap_width; /* :1130-1135 */
    const float discon_threshold = 3.0;
    for (int v = 0; v < H; v++) {
        int count = 0;
        for (int u = 0; u < W; u++) {
            if (D[(size_t)v * W + u] >= 0) {
                if (count >= 1 && count <= gw) {
                    int u_first = u - count, u_last = u - 1;
                    if (u_first > 0 && u_last < W - 1) {
                        float d1 = D[(size_t)v * W + u_first - 1], d2 = D[(size_t)v * W + u_last + 1];
                        float d_ipol = fabs(d1 - d2) < discon_threshold ? (d1 + d2) / 2 : std::min(d1, d2);
                        for (int uc = u_first; uc <= u_last; uc++) D[(size_t)v * W + uc] = d_ipol;
                    }
                }
                count = 0;
            } else
                count++;
        }
        if (P.add_corners) {
            for (int u = 0; u < W; u++)
                if (D[(size_t)v * W + u] >= 0) {
                    for (int u2 = std::max(u - gw, 0); u2 < u; u2++) D[(size_t)v * W + u2] = D[(size_t)v * W + u];
                    break;
                }
            for (int u = W - 1; u >= 0; u--)
                if (D[(size_t)v * W + u] >= 0) {
                    for (int u2 = u; u2 <= std::min(u + gw, W - 1); u2++) D[(size_t)v * W + u2] = D[(size_t)v * W + u];
                    break;
                }
        }
    }
    for (int u = 0; u < W; u++) {
        int count = 0;
        for (int v = 0; v < H; v++) {
            if (D[(size_t)v * W + u] >= 0) {
                if (count >= 1 && count <= gw) {
                    int v_first = v - count, v_last = v - 1;
                    if (v_first > 0 && v_last < H - 1) {
                        float d1 = D[(size_t)(v_first - 1) * W + u], d2 = D[(size_t)(v_last + 1) * W + u];
                        float d_ipol = fabs(d1 - d2) < discon_threshold ? (d1 + d2) / 2 : std::min(d1, d2);
                        for (int vc = v_first; vc <= v_last; vc++) D[(size_t)vc * W + u] = d_ipol;
                    }
                }
                count = 0;
            } else
                count++;
        }
        if (P.add_corners) {
            for (int v = 0; v < H; v++)
                if (D[(size_t)v * W + u] >= 0) {
                    for (int v2 = std::max(v - gw, 0); v2 < v; v2++) D[(size_t)v2 * W + u] = D[(size_t)v * W + u];
                    break;
                }
            for (int v = H - 1; v >= 0; v--)
                if (D[(size_t)v * W + u] >= 0) {
                    for (int v2 = v; v2 <= std::min(v + gw, H - 1); v2++) D[(size_t)v2 * W + u] = D[(size_t)v * W + u];
                    break;
                }
        }
    }
}

/* The reference's "absolute value" mask is `_mm_set1_ps(0x7FFFFFFF)` (elas.cpp:1329): the *integer* is converted
 * to float 2^31 = bits 0x4F000000, so the AND keeps only five exponent bits. */
inline float absq(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    b &= 0x4F000000u;
    memcpy(&x, &b, 4);
    return x;
}

/* one 8-tap step of elas.cpp:1413-1440: ring slots k = pixel index % 8; lane sums pair slot j with j+4, then
 * ((s0+s1)+s2)+s3 */
inline bool amean_tap(const float val[8], float val_curr, float &d_out) {
    float w[8], f[8];
    for (int k = 0; k < 8; k++) {
        float t = 4.0f - absq(val[k] - val_curr);
        w[k] = 0.0f > t ? 0.0f : t; /* _mm_max_ps(xconst0, t) */
        f[k] = val[k] * w[k];
    }
    float ws[4], fs[4];
    for (int j = 0; j < 4; j++) {
        ws[j] = w[j] + w[j + 4];
        fs[j] = f[j] + f[j + 4];
    }
    float weight_sum = ws[0] + ws[1] + ws[2] + ws[3];
    float factor_sum = fs[0] + fs[1] + fs[2] + fs[3];
    if (weight_sum > 0) {
        float d = factor_sum / weight_sum;
        if (d >= 0) {
            d_out = d;
            return true;
        }
    }
    return false;
}

/* elas.cpp:1297-1494, full-resolution branch :1400-1485; D_tmp canonical zero where the reference leaves it
 * uninitialised (:1308). */
/* one 4-tap step of the half-resolution branch (elas.cpp:1344-1362): plain left-to-right sums of the four slots */
inline bool amean_tap4(const float val[8], float val_curr, float &d_out) {
    float w[4], f[4];
    for (int k = 0; k < 4; k++) {
        float t = 4.0f - absq(val[k] - val_curr);
        w[k] = 0.0f > t ? 0.0f : t;
        f[k] = val[k] * w[k];
    }
    float weight_sum = w[0] + w[1] + w[2] + w[3];
    float factor_sum = f[0] + f[1] + f[2] + f[3];
    if (weight_sum > 0) {
        float d = factor_sum / weight_sum;
        if (d >= 0) {
            d_out = d;
            return true;
        }
    }
    return false;
}

void adaptive_mean(float *D, int W, int H, bool sub = false) {
    std::vector<float> C(D, D + (size_t)W * H), T((size_t)W * H, 0.0f);
    for (size_t i = 0; i < (size_t)W * H; i++)
        if (D[i] < 0) {
            C[i] = -10;
            T[i] = -10;
        }
    float val[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* the ring persists across rows and passes, as in the reference (:1324) */
    if (sub) { /* elas.cpp:1332-1397: 4-pixel window u-3..u around the centre u-1 */
        for (int v = 3; v < H - 3; v++) {
            for (int u = 0; u < 3; u++) val[u] = C[(size_t)v * W + u];
            for (int u = 3; u < W; u++) {
                float val_curr = C[(size_t)v * W + (u - 1)];
                val[u % 4] = C[(size_t)v * W + u];
                float d;
                if (amean_tap4(val, val_curr, d))
                    T[(size_t)v * W + (u - 1)] = d;
            }
        }
        for (int u = 3; u < W - 3; u++) {
            for (int v = 0; v < 3; v++) val[v] = T[(size_t)v * W + u];
            for (int v = 3; v < H; v++) {
                float val_curr = T[(size_t)(v - 1) * W + u];
                val[v % 4] = T[(size_t)v * W + u];
                float d;
                if (amean_tap4(val, val_curr, d))
                    D[(size_t)(v - 1) * W + u] = d;
            }
        }
        return;
    }
    for (int v = 3; v < H - 3; v++) {
        for (int u = 0; u < 7; u++) val[u] = C[(size_t)v * W + u];
        for (int u = 7; u < W; u++) {
            float val_curr = C[(size_t)v * W + (u - 3)];
            val[u % 8] = C[(size_t)v * W + u];
            float d;
            if (amean_tap(val, val_curr, d))
                T[(size_t)v * W + (u - 3)] = d;
        }
    }
    for (int u = 3; u < W - 3; u++) {
        for (int v = 0; v < 7; v++) val[v] = T[(size_t)v * W + u];
        for (int v = 7; v < H; v++) {
            float val_curr = T[(size_t)(v - 3) * W + u];
            val[v % 8] = T[(size_t)v * W + u];
            float d;
            if (amean_tap(val, val_curr, d))
                D[(size_t)(v - 3) * W + u] = d;
        }
    }
}

inline float median7_insertion(const float *src, int stride) { /* insertion sort of elas.cpp:1519-1528 */
    float vals[7];
    for (int j = 0; j < 7; j++) {
        float temp = src[(long)j * stride];
        int i = j - 1;
        while (i >= 0 && vals[i] > temp) {
            vals[i + 1] = vals[i];
            i--;
        }
        vals[i + 1] = temp;
    }
    return vals[3];
}

/* elas.cpp:1496-1560 */
void median(float *D, int W, int H) {
    std::vector<float> T((size_t)W * H, 0.0f);
    for (int u = 3; u < W - 3; u++)
        for (int v = 3; v < H - 3; v++) {
            size_t a = (size_t)v * W + u;
            T[a] = D[a] >= 0 ? median7_insertion(D + a - 3, 1) : D[a];
        }
    for (int u = 3; u < W - 3; u++)
        for (int v = 3; v < H - 3; v++) {
            size_t a = (size_t)v * W + u;
            if (D[a] >= 0)
                D[a] = median7_insertion(T.data() + a - (size_t)3 * W, W);
        }
}

/* ------------------------------------------------------------------------------------------------
 * Pipeline (elas.cpp:31-150)
 * ---------------------------------------------------------------------------------------------- */

std::map<std::string, std::vector<uint8_t>> g_store;

template <class T>
void put(const char *name, const T *data, size_t count) {
    std::vector<uint8_t> &v = g_store[name];
    v.resize(count * sizeof(T));
    if (count)
        memcpy(v.data(), data, count * sizeof(T));
}

int run(const elas_params &P, const uint8_t *I1, const uint8_t *I2, int W, int H, int stride, float *D1, float *D2, bool keep) {
    const size_t NI = (size_t)W * H;
    const bool sub = P.subsampling != 0;
    const int Wm = sub ? W / 2 : W, Hm = sub ? H / 2 : H; /* map size: elas.h:160-161 */
    const size_t N = (size_t)Wm * Hm;
    std::vector<uint8_t> desc1(NI * 16), desc2(NI * 16);
    descriptor(I1, W, H, stride, desc1.data(), sub);
    descriptor(I2, W, H, stride, desc2.data(), sub);
    int Wc, Hc;
    candidate_dims(P, W, H, Wc, Hc);
    std::vector<int16_t> dcan((size_t)Wc * Hc);
    support_raw(P, desc1.data(), desc2.data(), W, H, dcan.data());
    if (keep) {
        put("desc1", desc1.data(), desc1.size());
        put("desc2", desc2.data(), desc2.size());
        put("dcan_raw", dcan.data(), dcan.size());
        int32_t dd[2] = {Wc, Hc};
        put("dcan_dims", dd, 2);
    }
    std::vector<Pt> s = support_filter(P, dcan.data(), W, H);
    if (keep)
        put("support", (const int32_t *)s.data(), s.size() * 3);
    if (s.size() < 3)
        return (int)s.size(); /* elas.cpp:64-69: outputs untouched */
    const int32_t *sp = (const int32_t *)s.data();
    std::vector<int32_t> tri1 = delaunay(s, false), tri2 = delaunay(s, true);
    int nt1 = (int)tri1.size() / 3, nt2 = (int)tri2.size() / 3;
    std::vector<float> pl1((size_t)nt1 * 6), pl2((size_t)nt2 * 6);
    planes(sp, tri1.data(), nt1, pl1.data());
    planes(sp, tri2.data(), nt2, pl2.data());
    int32_t gd[3];
    grid_dims_of(P, W, H, gd);
    std::vector<int32_t> g1((size_t)gd[0] * gd[1] * gd[2]), g2(g1.size());
    grid(P, sp, (int)s.size(), W, H, false, g1.data());
    grid(P, sp, (int)s.size(), W, H, true, g2.data());
    if (keep) {
        put("tri1", tri1.data(), tri1.size());
        put("tri2", tri2.data(), tri2.size());
        put("planes1", pl1.data(), pl1.size());
        put("planes2", pl2.data(), pl2.size());
        put("grid1", g1.data(), g1.size());
        put("grid2", g2.data(), g2.size());
        put("grid_dims", gd, 3);
    }
    dense(P, sp, tri1.data(), pl1.data(), nt1, g1.data(), desc1.data(), desc2.data(), W, H, false, D1);
    dense(P, sp, tri2.data(), pl2.data(), nt2, g2.data(), desc1.data(), desc2.data(), W, H, true, D2);
    if (keep) {
        put("wta1", D1, N);
        put("wta2", D2, N);
    }
    lr_check(P, D1, D2, Wm, Hm);
    if (keep) {
        put("lr1", D1, N);
        put("lr2", D2, N);
    }
    speckle(P, D1, Wm, Hm);
    if (!P.postprocess_only_left)
        speckle(P, D2, Wm, Hm);
    if (keep) {
        put("speckle1", D1, N);
        put("speckle2", D2, N);
    }
    gap(P, D1, Wm, Hm);
    if (!P.postprocess_only_left)
        gap(P, D2, Wm, Hm);
    if (keep) {
        put("gap1", D1, N);
        put("gap2", D2, N);
    }
    if (P.filter_adaptive_mean) {
        adaptive_mean(D1, Wm, Hm, sub);
        if (!P.postprocess_only_left)
            adaptive_mean(D2, Wm, Hm, sub);
    }
    if (keep) {
        put("amean1", D1, N);
        put("amean2", D2, N);
    }
    if (P.filter_median) {
        median(D1, Wm, Hm);
        if (!P.postprocess_only_left)
            median(D2, Wm, Hm);
    }
    if (keep) {
        put("final1", D1, N);
        put("final2", D2, N);
    }
    return (int)s.size();
}

}  // namespace

extern "C" {

double orc_process(const elas_params *p, const uint8_t *I1, const uint8_t *I2, int W, int H, int stride, float *D1, float *D2, int canonical, int reps) {
    (void)canonical; /* always canonical: the restatement has no uninitialised reads */
    if (reps < 1)
        reps = 1;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) run(*p, I1, I2, W, H, stride, D1, D2, false);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / reps;
}

int orc_run_stages(const elas_params *p, const uint8_t *I1, const uint8_t *I2, int W, int H, int stride) {
    g_store.clear();
    std::vector<float> D1((size_t)W * H, 0.f), D2((size_t)W * H, 0.f); /* driver passes zeroed maps (stereo_vision.cpp:304-305) */
    return run(*p, I1, I2, W, H, stride, D1.data(), D2.data(), true);
}

long orc_size(const char *name) {
    auto it = g_store.find(name);
    return it == g_store.end() ? -1 : (long)it->second.size();
}

long orc_get(const char *name, void *out, long cap) {
    auto it = g_store.find(name);
    if (it == g_store.end())
        return -1;
    long n = (long)it->second.size();
    if (n > cap)
        return -2;
    memcpy(out, it->second.data(), n);
    return n;
}

void orc_descriptor(const uint8_t *I, int W, int H, int stride, uint8_t *desc) { descriptor(I, W, H, stride, desc); }

void orc_support_raw(const elas_params *p, const uint8_t *desc1, const uint8_t *desc2, int W, int H, int16_t *dcan) {
    support_raw(*p, desc1, desc2, W, H, dcan);
}

int orc_support_filter(const elas_params *p, int16_t *dcan, int W, int H, int32_t *support, int cap) {
    std::vector<Pt> s = support_filter(*p, dcan, W, H);
    int n = (int)s.size();
    if (n > cap)
        return -n;
    if (n)
        memcpy(support, s.data(), sizeof(Pt) * n);
    return n;
}

int orc_delaunay(const float *xy, int n, int32_t *tri_out, int cap) {
    std::vector<int32_t> tri;
    int nt = dc::triangulate(xy, n, tri);
    for (int i = 0; i < nt * 3 && i < cap * 3; i++) tri_out[i] = tri[i];
    return nt;
}

void orc_planes(const int32_t *support, const int32_t *tri, int nt, float *planes_out) { planes(support, tri, nt, planes_out); }

void orc_grid(const elas_params *p, const int32_t *support, int n, int W, int H, int right_image, int32_t *grid_out) {
    grid(*p, support, n, W, H, right_image != 0, grid_out);
}

void orc_dense(const elas_params *p, const int32_t *support, const int32_t *tri, const float *planes_in, int nt, const int32_t *grid_in,
               const uint8_t *desc1, const uint8_t *desc2, int W, int H, int right_image, float *D) {
    dense(*p, support, tri, planes_in, nt, grid_in, desc1, desc2, W, H, right_image != 0, D);
}

void orc_lr_check(const elas_params *p, float *D1, float *D2, int W, int H) { lr_check(*p, D1, D2, W, H); }
void orc_speckle(const elas_params *p, float *D, int W, int H) { speckle(*p, D, W, H); }
void orc_gap(const elas_params *p, float *D, int W, int H) { gap(*p, D, W, H); }
void orc_adaptive_mean(float *D, int W, int H) { adaptive_mean(D, W, H); }
void orc_adaptive_mean_sub(float *D, int W, int H) { adaptive_mean(D, W, H, true); }
void orc_median(float *D, int W, int H) { median(D, W, H); }

} /* extern "C" */
