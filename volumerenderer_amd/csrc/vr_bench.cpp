/*
 * vr_bench.cpp -- command-line driver with the shape of the reference's main (src/main.zig:11-164):
 * load a .vol dataset, build a SlimKDT per partition heuristic, render the iso sweep, time
 * renderImage per frame, write renders/out_iso{d}.png, print the reference's comparison table
 * (ms/frame, layers, deviating partitions, nodes checked [mil], time per frame vs balanced).
 * Host side of the library above its C ABI; the rendering itself runs on the GPU.
 *
 *   vr_bench [--volume C60.vol] [--size 512] [--iso 0:256] [--heuristics main|center|all]
 *            [--out DIR | --no-images] [--sweep] [--device 0] [--viewport 2.0] [--camera x,y,z]
 *            [--gpus N [--shard iso|rows]] [--gpu-build]
 * --gpu-build: trees are built on the GPU (vr_upload_volume + vr_build_kdt) instead of by the host builder.
 * --gpus N (implies --sweep): N contexts, on devices 0..N-1 as far as the node has them (beyond that the
 * devices are used again), driven by this one thread through vr_mgpu_render_sweep.
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vr.h"

static const struct { int id; const char *name; } kMainList[] = {
    /* the list of src/main.zig:60-75 */
    { VR_H_ALWAYS_CENTER, "AlwaysCenter" },
    { VR_H_FLEXIBLE_FIRST5, "FlexibleFirst5" },
    { VR_H_FLEXIBLE_AFTER15, "FlexibleAfter15" },
    { VR_H_FLEXIBLE_AFTER20, "FlexibleAfter20" },
    { VR_H_FLEXIBLE_CENTER_WEIGHTED30, "FlexibleCenterWeigthed30" },
    { VR_H_FLEXIBLE_CENTER_WEIGHTED50, "FlexibleCenterWeigthed50" },
    { VR_H_FLEXIBLE_CENTER_WEIGHTED75, "FlexibleCenterWeigthed75" },
};

static const char *kAllNames[VR_H_COUNT] = {
    "AlwaysCenter", "MinimizeSpan", "FlexibleFirst5", "FlexibleFirst10", "FlexibleAfter10", "FlexibleAfter15",
    "FlexibleAfter20", "FlexibleCenterWeigthed5", "FlexibleCenterWeigthed10", "FlexibleCenterWeigthed30",
    "FlexibleCenterWeigthed50", "FlexibleCenterWeigthed75", "FlexibleCenterWeigthed100",
};

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc__ = (call);                                                                 \
        if (rc__ != VR_OK) {                                                               \
            std::fprintf(stderr, "%s failed: %d (%s)\n", #call, rc__, vr_strerror(rc__)); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

int main(int argc, char **argv)
{
    std::string volume = "C60.vol", out_dir = "renders", heur = "main";
    uint32_t size = 512, iso_begin = 0, iso_end = 256;
    int device = 0;
    bool images = true;
    bool sweep = false; /* the whole iso loop in one vr_render_sweep call (frames into page-locked memory) */
    int n_gpus = 1, shard = VR_SHARD_ISO;
    bool gpu_build = false;
    float viewport = 2.0f, cam[3] = { -30.0f, -300.0f, -30.0f }; /* src/main.zig:37,126 */
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--volume") volume = next();
        else if (a == "--size") size = (uint32_t)std::atoi(next());
        else if (a == "--iso") { if (std::sscanf(next(), "%u:%u", &iso_begin, &iso_end) != 2) { std::fprintf(stderr, "--iso a:b\n"); return 2; } }
        else if (a == "--heuristics") heur = next();
        else if (a == "--out") out_dir = next();
        else if (a == "--no-images") images = false;
        else if (a == "--sweep") sweep = true;
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--gpu-build") gpu_build = true;
        else if (a == "--gpus") { n_gpus = std::atoi(next()); sweep = true; }
        else if (a == "--shard") { std::string m = next(); shard = m == "rows" ? VR_SHARD_ROWS : VR_SHARD_ISO; }
        else if (a == "--viewport") viewport = (float)std::atof(next());
        else if (a == "--camera") { if (std::sscanf(next(), "%f,%f,%f", &cam[0], &cam[1], &cam[2]) != 3) { std::fprintf(stderr, "--camera x,y,z\n"); return 2; } }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (n_gpus < 1 || n_gpus > 64) { std::fprintf(stderr, "--gpus 1..64\n"); return 2; }
    if (iso_end > 256 || iso_begin >= iso_end || size == 0) { std::fprintf(stderr, "bad --iso / --size\n"); return 2; }

    std::printf("Loading dataset: '%s'\n", volume.c_str());
    uint8_t *vox = nullptr;
    uint32_t res[3];
    CHECK(vr_load_ucd(volume.c_str(), &vox, res));
    std::printf("init data: %ux%ux%u\n", res[0], res[1], res[2]);

    vr_view view;
    std::memcpy(view.src, cam, sizeof cam);
    vr_main_camera(res, cam, view.forw, view.up);
    view.viewport_width = viewport;
    view.width = view.height = size;

    std::vector<std::pair<int, const char *>> list;
    if (heur == "main") for (auto &h : kMainList) list.push_back({ h.id, h.name });
    else if (heur == "center") list.push_back({ VR_H_ALWAYS_CENTER, "AlwaysCenter" });
    else if (heur == "all") for (int h = 0; h < VR_H_COUNT; h++) list.push_back({ h, kAllNames[h] });
    else { std::fprintf(stderr, "--heuristics main|center|all\n"); return 2; }

    vr_ctx *ctx = nullptr;
    CHECK(vr_ctx_create(device, &ctx));
    std::vector<vr_ctx *> group(1, ctx);
    for (int g = 1, n_dev = 0; g < n_gpus; g++) {
        vr_ctx *more = nullptr;
        int rc = n_dev ? VR_EARG : vr_ctx_create(device + g, &more);
        if (rc == VR_EARG) { /* no such device: go round the ones there are */
            if (!n_dev) n_dev = g;
            rc = vr_ctx_create(device + g % n_dev, &more);
        }
        CHECK(rc);
        group.push_back(more);
    }
    std::vector<uint8_t> img((size_t)size * size * 3u);
    double tot_time_balanced = 0.0;
    const uint32_t n_iso = iso_end - iso_begin;
    for (size_t hi = 0; hi < list.size(); hi++) {
        std::printf("\n\nBUILDING KDT WITH HEURISTIC %s\n\n", list[hi].second);
        uint32_t *nodes = nullptr;
        uint64_t n_slots = 0;
        uint8_t root_dim[3];
        vr_build_stats bs;
        int rc = VR_OK;
        if (gpu_build) {
            for (size_t g = 0; g < group.size() && rc == VR_OK; g++) {
                rc = vr_upload_volume(group[g], vox, res[0], res[1], res[2]);
                if (rc == VR_OK) rc = vr_build_kdt(group[g], list[hi].first, &bs);
            }
        } else {
            rc = vr_host_build_kdt(vox, res, list[hi].first, (uint64_t)1 << 30, &nodes, &n_slots, root_dim, &bs);
            for (size_t g = 0; g < group.size() && rc == VR_OK; g++) rc = vr_upload_kdt(group[g], nodes, n_slots, root_dim);
        }
        if (rc != VR_OK) { /* src/main.zig:100-104 */
            std::printf("\n FAILD TO BUILD KDT! (%s)\n", vr_strerror(rc));
            vr_free(nodes);
            continue;
        }
        vr_free(nodes);
        std::printf("\n KDT Built\n nodes: %llu\n non center partitions: %llu\n layers: %llu (deepest point)\n",
                    (unsigned long long)bs.number_of_nodes, (unsigned long long)bs.non_center_partition_count,
                    (unsigned long long)bs.layer_count);
        double tot_ms = 0.0;
        uint64_t node_checks = 0;
        if (sweep) {
            /* src/main.zig:111-138 as one call: a ray is traced once per interval of iso values that cannot change
             * the outcome of a range test on its way; every frame and counter equals the per-iso call's */
            const size_t frame = (size_t)size * size * 3u;
            void *frames = nullptr;
            CHECK(vr_alloc_host(frame * n_iso, &frames));
            std::vector<vr_render_stats> st(n_iso);
            const auto t0 = std::chrono::steady_clock::now();
            if (group.size() > 1)
                CHECK(vr_mgpu_render_sweep(group.data(), (uint32_t)group.size(), &view, iso_begin, n_iso, shard, (uint8_t *)frames, st.data()));
            else
                CHECK(vr_render_sweep(ctx, &view, iso_begin, n_iso, (uint8_t *)frames, st.data(), nullptr));
            const auto t1 = std::chrono::steady_clock::now();
            tot_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
            for (uint32_t k = 0; k < n_iso; k++) {
                node_checks += st[k].nodes_checked;
                if (images) {
                    char name[512];
                    std::snprintf(name, sizeof name, "%s/out_iso%u.png", out_dir.c_str(), iso_begin + k);
                    CHECK(vr_write_png_rgb24(name, size, size, (const uint8_t *)frames + frame * k));
                }
            }
            vr_free_host(frames);
        } else
        for (uint32_t iso = iso_begin; iso < iso_end; iso++) {
            vr_render_stats st;
            const auto t0 = std::chrono::steady_clock::now();
            CHECK(vr_render(ctx, &view, (uint8_t)iso, img.data(), &st, nullptr)); /* timed like src/main.zig:119-135 */
            const auto t1 = std::chrono::steady_clock::now();
            tot_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
            node_checks += st.nodes_checked;
            if (images) {
                char name[512];
                std::snprintf(name, sizeof name, "%s/out_iso%u.png", out_dir.c_str(), iso);
                CHECK(vr_write_png_rgb24(name, size, size, img.data()));
            }
        }
        if (hi == 0) tot_time_balanced = tot_ms;
        std::printf("node checks: %llu  (%llu million)\n", (unsigned long long)node_checks, (unsigned long long)(node_checks / 1000000ull));
        std::printf("Render time per img: %.4fms \n", tot_ms / n_iso);
        std::printf("\n\n(ms/frame)\t(layers)\t(deviating partitions)\t(nodes checked [mil]) (time per frame vs balanced)\n"
                    "      %.4f ,            %llu ,                  %llu ,            %llu ,                 %.4f\n\n\n",
                    tot_ms / n_iso, (unsigned long long)bs.layer_count, (unsigned long long)bs.non_center_partition_count,
                    (unsigned long long)(node_checks / 1000000ull), tot_time_balanced > 0 ? tot_ms / tot_time_balanced : 0.0);
        std::printf("Mrays/s (host-timed, frames copied to the host): %.1f\n", (double)size * size * n_iso / (tot_ms * 1e-3) / 1e6);
    }
    for (vr_ctx *c : group) vr_ctx_destroy(c);
    vr_free(vox);
    return 0;
}
