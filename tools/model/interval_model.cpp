// Host-side model: for a sample of rays, how many DISTINCT traversals does the 0..255 iso sweep contain?
// A traversal depends on the iso value only through the outcomes of the tests iso in [min, max] at the
// nodes it visits, so it is identical for every iso value in the intersection of the intervals those
// outcomes allow.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vr.h"
#include "vr_trace_core.h"

struct IntervalFetch {
    const uint32_t *nodes;
    uint32_t iso;
    mutable uint32_t *a, *b;
    mutable uint64_t *fetches;
    uint32_t operator()(uint32_t g) const
    {
        const uint32_t n = nodes[g], mn = n & 0xffu, mx = (n >> 8) & 0xffu;
        (*fetches)++;
        if (iso < mn) { if (mn - 1 < *b) *b = mn - 1; }
        else if (iso > mx) { if (mx + 1 > *a) *a = mx + 1; }
        else { if (mn > *a) *a = mn; if (mx < *b) *b = mx; }
        return n;
    }
};
struct HostStack {
    uint32_t word[40];
    int cur = 1;
    void reset() { cur = 1; }
    void push(uint32_t w) { word[cur++] = w; }
    void up() { cur--; }
    uint32_t peek_up(uint32_t k) const { return word[cur - (int)k]; }
    uint32_t depth() const { return (uint32_t)(cur - 1); }
};

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "Skull.vol";
    const uint32_t W = argc > 2 ? atoi(argv[2]) : 1024, step = argc > 3 ? atoi(argv[3]) : 8;
    uint8_t *vox; uint32_t res[3];
    if (vr_load_ucd(path, &vox, res)) return 1;
    uint32_t *nodes; uint64_t n_slots; uint8_t rd[3];
    if (vr_host_build_kdt(vox, res, 0, 0, &nodes, &n_slots, rd, nullptr)) return 2;
    std::vector<uint32_t> prep(vr_prepared_words(n_slots));
    uint32_t depth;
    if (vr_host_prepare_kdt(nodes, n_slots, rd, prep.data(), prep.size(), &depth, nullptr)) return 3;
    float pos[3] = { -30, -300, -30 }, forw[3], up[3];
    vr_main_camera(res, pos, forw, up);
    VrViewK k;
    k.ox = pos[0]; k.oy = pos[1]; k.oz = pos[2]; k.fx = forw[0]; k.fy = forw[1]; k.fz = forw[2];
    k.ux = up[0]; k.uy = up[1]; k.uz = up[2];
    k.rx = k.fy * k.uz - k.fz * k.uy; k.ry = k.fz * k.ux - k.fx * k.uz; k.rz = k.fx * k.uy - k.fy * k.ux;
    k.pixel_size = 2.0f / (float)W; k.half_w = (float)(W / 2); k.half_h = (float)(W / 2);
    k.width = W; k.height = W; k.rdx = rd[0]; k.rdy = rd[1]; k.rdz = rd[2];
    k.root_hi = vr_pack10(rd[0], rd[1], rd[2]); k.n_slots = n_slots;
    uint64_t rays = 0, traces = 0, fetch_all = 0, fetch_distinct = 0, checked_all = 0, checked_distinct = 0;
    uint64_t hist[257] = { 0 };
    for (uint32_t y = 0; y < W; y += step)
        for (uint32_t x = 0; x < W; x += step) {
            VrRay ray = vr_make_ray(k, x, y);
            if (ray.oob) continue;
            rays++;
            uint32_t n_tr = 0;
            for (uint32_t iso = 0; iso < 256;) {
                HostStack st; VrLane L; uint32_t esc = 0, a = 0, b = 255;
                uint64_t f = 0;
                vr_lane_begin(L, st, k, ray.dx, ray.dy, ray.dz, ray.t);
                IntervalFetch fe{ prep.data(), iso, &a, &b, &f };
                for (;;) {
                    const uint32_t go = vr_lane_descend(L, iso, fe, st, 64);
                    if (go == VR_GO_HIT) break;
                    if (!vr_lane_move(L, k, esc)) break;
                    while (!vr_lane_pop(L, st, 64)) {}
                }
                n_tr++;
                fetch_distinct += f; checked_distinct += L.checked;
                fetch_all += f * (b - iso + 1); checked_all += (uint64_t)L.checked * (b - iso + 1);
                iso = b + 1;
            }
            traces += n_tr;
            hist[n_tr]++;
        }
    printf("rays %llu  distinct traversals per ray %.2f of 256 (%.1fx fewer)\n", (unsigned long long)rays,
           (double)traces / rays, 256.0 * rays / traces);
    printf("node fetches: all isos %llu, distinct only %llu (%.1fx fewer); reference checks: all %llu, distinct %llu (%.1fx)\n",
           (unsigned long long)fetch_all, (unsigned long long)fetch_distinct, (double)fetch_all / fetch_distinct,
           (unsigned long long)checked_all, (unsigned long long)checked_distinct, (double)checked_all / checked_distinct);
    printf("histogram of traversals per ray:");
    for (int i = 0; i <= 256; i++) if (hist[i]) printf(" %d:%llu", i, (unsigned long long)hist[i]);
    printf("\n");
    return 0;
}
