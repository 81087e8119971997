// Host-side model of the traversal's memory behaviour: counts, for a sample of rays, how many node
// fetches the resume traversal issues and how many of them a layout that keeps both children's density
// ranges in the parent (and the sibling's range in the stack entry) would still need.
// Build: g++ -O2 -I../../volumerenderer_amd/csrc -I../../include fetch_model.cpp -L../../volumerenderer_amd -lvr_hip ...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "vr.h"
#include "vr_trace_core.h"

struct CountFetch {
    const uint32_t *nodes;
    mutable uint64_t *fetches, *inner_pass;
    uint32_t iso;
    uint32_t operator()(uint32_t g) const
    {
        const uint32_t n = nodes[g];
        (*fetches)++;
        if (vr_node_holds(n, iso) && !(n & VR_N_LEAF)) (*inner_pass)++;
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
    const uint32_t W = argc > 2 ? atoi(argv[2]) : 1024, step = argc > 3 ? atoi(argv[3]) : 4;
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
    uint64_t fetches = 0, inner = 0, moves = 0, rays = 0, checked = 0;
    uint64_t hist[32] = { 0 };
    for (uint32_t iso = 4; iso < 256; iso += 32)
        for (uint32_t y = 0; y < W; y += step)
            for (uint32_t x = 0; x < W; x += step) {
                VrRay ray = vr_make_ray(k, x, y);
                if (ray.oob) continue;
                rays++;
                HostStack st; VrLane L; uint32_t esc = 0;
                vr_lane_begin(L, st, k, ray.dx, ray.dy, ray.dz, ray.t);
                CountFetch f{ prep.data(), &fetches, &inner, iso };
                for (;;) {
                    const uint64_t f0 = fetches;
                    const uint32_t go = vr_lane_descend(L, iso, f, st, 64);
                    const uint64_t d = fetches - f0;
                    hist[d < 31 ? d : 31]++;
                    if (go == VR_GO_HIT) break;
                    moves++;
                    if (!vr_lane_move(L, k, esc)) break;
                    while (!vr_lane_pop(L, st, 64)) {}
                }
                checked += L.checked;
            }
    printf("rays %llu  moves %llu  reference checks %llu  fetches now %llu  (%.2f per descent)  inner-pass fetches %llu (%.1f%% of now)\n",
           (unsigned long long)rays, (unsigned long long)moves, (unsigned long long)checked, (unsigned long long)fetches,
           (double)fetches / (double)(moves + rays), (unsigned long long)inner, 100.0 * inner / fetches);
    uint64_t tot = 0; for (int i = 0; i < 32; i++) tot += hist[i];
    printf("fetches per descent histogram (%%):");
    for (int i = 1; i < 16; i++) printf(" %d:%.1f", i, 100.0 * hist[i] / tot);
    printf("\n");
    return 0;
}
