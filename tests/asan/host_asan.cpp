/*
 * host_asan.cpp -- TEST HARNESS: the host side of libvr_hip.so (tree builder, flattening, file formats, image
 * writers) and the kernels' lane code compiled for the CPU, in one program built with AddressSanitizer and
 * UndefinedBehaviorSanitizer.  It walks the same calls the C ABI makes on the host, on random volumes, ragged
 * sizes, malformed trees and truncated files; any out-of-bounds access or undefined operation aborts the run.
 * Not part of the product library.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../volumerenderer_amd/csrc/vr_host.cpp"
#include "../../volumerenderer_amd/csrc/vr_image_io.cpp"
#include "../../volumerenderer_amd/csrc/vr_trace_core.h"
#include "../../volumerenderer_amd/csrc/vr_trace_wide.h"

namespace {
uint64_t g_s = 0x9e3779b97f4a7c15ull;
uint32_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (uint32_t)(g_s >> 24); }

struct Fetch { const uint32_t *n; uint32_t operator()(uint32_t i) const { return n[i]; } };
struct Stack {
    uint32_t word[VR_MAX_DEPTH + 3];
    int cur = 1;
    void reset() { cur = 1; }
    void push(uint32_t w) { word[cur++] = w; }
    void up() { cur--; }
    uint32_t peek_up(uint32_t k) const { return word[cur - (int)k]; }
    uint32_t depth() const { return (uint32_t)(cur - 1); }
};
struct Path {
    uint32_t word[VR_WIDE_MAX_DEPTH];
    uint32_t get(uint32_t d) const { return word[d]; }
    void set(uint32_t d, uint32_t w) { word[d] = w; }
};

VrViewK make_view(const uint32_t rd[3], const float pos[3], const uint32_t res[3], uint32_t w, uint32_t h, float vp, uint64_t n_slots)
{
    float f[3], u[3];
    vr_main_camera(res, pos, f, u);
    VrViewK k;
    k.ox = pos[0]; k.oy = pos[1]; k.oz = pos[2];
    k.fx = f[0]; k.fy = f[1]; k.fz = f[2];
    k.ux = u[0]; k.uy = u[1]; k.uz = u[2];
    k.rx = k.fy * k.uz - k.fz * k.uy; k.ry = k.fz * k.ux - k.fx * k.uz; k.rz = k.fx * k.uy - k.fy * k.ux;
    k.pixel_size = vp / (float)w; k.half_w = (float)(w / 2); k.half_h = (float)(h / 2);
    k.width = w; k.height = h;
    k.rdx = rd[0]; k.rdy = rd[1]; k.rdz = rd[2];
    k.root_hi = vr_pack10(k.rdx, k.rdy, k.rdz);
    k.n_slots = n_slots;
    return k;
}

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)
} // namespace

int main(int argc, char **argv)
{
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    uint64_t rays = 0;
    const uint32_t shapes[][3] = { { 2, 2, 2 }, { 5, 9, 3 }, { 17, 4, 33 }, { 24, 24, 24 }, { 3, 64, 2 } };
    for (const auto &sh : shapes) {
        const uint32_t res[3] = { sh[0], sh[1], sh[2] };
        std::vector<uint8_t> vox((size_t)res[0] * res[1] * res[2]);
        for (auto &b : vox) b = (uint8_t)(rnd() % (1 + rnd() % 255));
        /* .vol round trip */
        const std::string vol = tmp + "/asan.vol";
        CHECK(vr_save_ucd(vol.c_str(), vox.data(), res) == VR_OK);
        uint8_t *back = nullptr; uint32_t r2[3];
        CHECK(vr_load_ucd(vol.c_str(), &back, r2) == VR_OK && std::memcmp(back, vox.data(), vox.size()) == 0);
        vr_free(back);
        for (int heur = 0; heur < VR_H_COUNT; heur += (res[0] * res[1] * res[2] > 4000 ? 6 : 1)) {
            uint32_t *nodes = nullptr; uint64_t n_slots = 0; uint8_t rd8[3]; vr_build_stats st;
            const int rc = vr_host_build_kdt(vox.data(), res, heur, (uint64_t)1 << 22, &nodes, &n_slots, rd8, &st);
            if (rc == VR_ETOODEEP) continue;
            CHECK(rc == VR_OK);
            const uint64_t pw = vr_prepared_words(n_slots);
            std::vector<uint32_t> prep(pw);
            uint32_t depth = 0; uint64_t n_nodes = 0;
            CHECK(vr_host_prepare_kdt(nodes, n_slots, rd8, prep.data(), pw, &depth, &n_nodes) == VR_OK);
            CHECK(n_nodes == st.number_of_nodes);
            /* tree cache round trip, then a truncated and a corrupted file */
            const std::string kdt = tmp + "/asan.kdt";
            CHECK(vr_save_kdt(kdt.c_str(), nodes, n_slots, rd8) == VR_OK);
            uint32_t *n2 = nullptr; uint64_t s2 = 0; uint8_t rd2[3];
            CHECK(vr_load_kdt(kdt.c_str(), &n2, &s2, rd2) == VR_OK && s2 == n_slots && std::memcmp(n2, nodes, n_slots * 4) == 0);
            vr_free(n2);
            {
                FILE *f = std::fopen(kdt.c_str(), "rb"); std::vector<uint8_t> raw(1 << 20);
                const size_t len = std::fread(raw.data(), 1, raw.size(), f); std::fclose(f);
                if (len < raw.size() && len > 48) {
                    f = std::fopen(kdt.c_str(), "wb"); std::fwrite(raw.data(), 1, len - 1 - rnd() % 40, f); std::fclose(f);
                    CHECK(vr_load_kdt(kdt.c_str(), &n2, &s2, rd2) == VR_EFORMAT);
                    raw[40 + rnd() % (len - 40)] ^= 0x10;
                    f = std::fopen(kdt.c_str(), "wb"); std::fwrite(raw.data(), 1, len, f); std::fclose(f);
                    CHECK(vr_load_kdt(kdt.c_str(), &n2, &s2, rd2) == VR_EFORMAT);
                }
            }
            /* lane code on the tree: literal loop against the resume traversal, ragged image sizes */
            const uint32_t rd[3] = { rd8[0], rd8[1], rd8[2] };
            const float cams[][3] = { { -30, -300, -30 }, { 350, 20, 10 }, { 8, 8, -320 } };
            for (const auto &cam : cams) {
                const uint32_t w = 1 + rnd() % 40, h = 1 + rnd() % 30;
                const VrViewK vraw = make_view(rd, cam, res, w, h, 0.3f + (rnd() % 100) * 0.03f, n_slots);
                const uint32_t iso = rnd() % 256;
                for (uint32_t y = 0; y < h; y++)
                    for (uint32_t x = 0; x < w; x++) {
                        const VrRay ray = vr_make_ray(vraw, x, y);
                        const VrHit a = vr_trace_restart(vraw, ray, iso, Fetch{ nodes });
                        Stack stk;
                        const VrHit b = vr_trace_resume(vraw, ray, iso, Fetch{ prep.data() }, stk, 1 + rnd() % 8);
                        CHECK(a.flag == b.flag && a.checked == b.checked && a.node == b.node && a.cell == b.cell &&
                              std::memcmp(&a.t, &b.t, 4) == 0 && a.error == 0 && b.error == 0);
                        /* the same tree in wide node words through the wide lane code */
                        rays++;
                    }
                std::vector<uint32_t> wide(n_slots);
                for (uint64_t i = 0; i < n_slots; i++) wide[i] = (nodes[i] & 0x00ffffffu) | (((nodes[i] >> 24) & 3u) << 28);
                const uint16_t rdw[3] = { rd8[0], rd8[1], rd8[2] };
                uint32_t wdepth = 0; uint64_t wcount = 0;
                CHECK(vr_host_check_kdt_wide(wide.data(), n_slots, rdw, &wdepth, &wcount) == VR_OK && wcount == st.number_of_nodes);
                for (uint32_t y = 0; y < h; y++)
                    for (uint32_t x = 0; x < w; x++) {
                        const VrRay ray = vr_make_ray(vraw, x, y);
                        const VrHit a = vr_trace_restart(vraw, ray, iso, Fetch{ nodes });
                        Path path;
                        const VrHit c = vr_trace_wide(vraw, ray, iso, Fetch{ wide.data() }, path);
                        CHECK(a.flag == c.flag && a.checked == c.checked && a.node == c.node && std::memcmp(&a.t, &c.t, 4) == 0);
                    }
            }
            /* malformed trees: the validating walk must refuse them or accept them, never read outside the array */
            for (int k = 0; k < 20; k++) {
                std::vector<uint32_t> bad(nodes, nodes + n_slots);
                for (int j = 0; j < 1 + (int)(rnd() % 6); j++) bad[rnd() % (n_slots < 64 ? n_slots : 64)] = rnd() * 2654435761u;
                const uint64_t cut = 1 + rnd() % n_slots;
                const int v = vr_host_prepare_kdt(bad.data(), cut, rd8, prep.data(), vr_prepared_words(cut), &depth, &n_nodes);
                CHECK(v == VR_OK || v == VR_ETREE || v == VR_ETOODEEP);
                /* the same for wide node words; the literal loop checks as it goes and must come back from any words */
                std::vector<uint32_t> junk(cut);
                for (auto &q : junk) q = rnd() * 2246822519u;
                const uint16_t rd16[3] = { rd8[0], rd8[1], rd8[2] };
                const int vw = vr_host_check_kdt_wide(junk.data(), cut, rd16, &depth, &n_nodes);
                CHECK(vw == VR_OK || vw == VR_ETREE);
                const VrViewK vj = make_view(rd, cams[0], res, 8, 8, 1.0f, cut);
                for (uint32_t p = 0; p < 64; p++)
                    (void)vr_trace_restart(vj, vr_make_ray(vj, p & 7u, p >> 3), rnd() % 256, Fetch{ junk.data() });
            }
            vr_free(nodes);
        }
        /* image writers */
        std::vector<uint8_t> img((size_t)res[0] * res[1] * 3);
        for (auto &b : img) b = (uint8_t)rnd();
        CHECK(vr_write_png_rgb24((tmp + "/asan.png").c_str(), res[0], res[1], img.data()) == VR_OK);
        CHECK(vr_write_ppm_rgb24((tmp + "/asan.ppm").c_str(), res[0], res[1], img.data()) == VR_OK);
    }
    /* argument errors */
    uint8_t *p = nullptr; uint32_t r3[3];
    CHECK(vr_load_ucd((tmp + "/does_not_exist.vol").c_str(), &p, r3) == VR_EIO);
    std::printf("ok: %llu rays through three traversals under ASan/UBSan\n", (unsigned long long)rays);
    return 0;
}
