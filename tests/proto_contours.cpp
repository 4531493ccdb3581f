// Host-side harness for the border-following logic shared with the gfx950 kernel (aruco_trace.hpp).
// TEST INFRASTRUCTURE: lets the CPU test-suite compare "independent read-only traces from candidate starts"
// against the oracle's sequential Suzuki-Abe scan without a GPU.  Not part of the product library.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../orb_slam2_aruco_amd/csrc/aruco_trace.hpp"

using namespace orbfe;

// ---- the two reductions of aruco_trace.hpp "FEWER WALKS", switchable so that the tests can run every formulation with and without
static int g_filter = 0, g_clean = 0;
extern "C" void proto_set_reductions(int filter_candidates, int clean_specks) { g_filter = filter_candidates; g_clean = clean_specks; }

// one pass of the speck window on a padded bit image of `prow` rows x wpr words (+ 2 spare words), zero beyond its frame
template <int WW, int HH>
static void speck_pass(std::vector<uint32_t>& bits, int wpr, int prow)
{
    const int pad = HH + 1, nr = prow + 2 * pad;   // rows -pad .. prow + pad - 1 of the zero-extended image
    auto P = [&](int r, int j) -> uint32_t { return (r >= 0 && r < prow && j >= 0 && j < wpr) ? bits[(size_t)r * wpr + j] : 0u; };
    std::vector<uint32_t> full((size_t)nr * wpr), side((size_t)nr * wpr), an((size_t)nr * wpr, 0u);
    for (int r = -pad; r < prow + pad; r++)
        for (int j = 0; j < wpr; j++) speck_row_masks<WW>(P(r, j), P(r, j + 1), &full[(size_t)(r + pad) * wpr + j], &side[(size_t)(r + pad) * wpr + j]);
    for (int r = -pad; r + HH + 1 < prow + pad; r++)   // anchor rows whose rim lies in the extended range (the others only see zero rows)
        for (int j = 0; j < wpr; j++) {
            uint32_t occ = full[(size_t)(r + pad) * wpr + j] | full[(size_t)(r + pad + HH + 1) * wpr + j];
            for (int k = 1; k <= HH; k++) occ |= side[(size_t)(r + pad + k) * wpr + j];
            an[(size_t)(r + pad) * wpr + j] = ~occ;
        }
    for (int r = 0; r < prow; r++)
        for (int j = 0; j < wpr; j++) {
            uint32_t e = 0, el = 0;
            for (int dy = 1; dy <= HH; dy++) {
                e |= an[(size_t)(r - dy + pad) * wpr + j];
                if (j) el |= an[(size_t)(r - dy + pad) * wpr + j - 1];
            }
            bits[(size_t)r * wpr + j] &= ~speck_dilate<WW>(e, el);
        }
}

static std::vector<uint32_t> padded_bits(const uint8_t* img, int w, int h)
{
    const int wpr = (w + 2 + 31) / 32;
    std::vector<uint32_t> bits((size_t)wpr * (h + 2) + 2, 0); // + spare words for ring8's funnel read
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            if (img[(size_t)y * w + x]) bits[(size_t)(y + 1) * wpr + ((x + 1) >> 5)] |= 1u << ((x + 1) & 31);
    if (g_clean) {
        speck_pass<ORBFE_SPECK_W1, ORBFE_SPECK_H1>(bits, wpr, h + 2);
        speck_pass<ORBFE_SPECK_W2, ORBFE_SPECK_H2>(bits, wpr, h + 2);
    }
    return bits;
}

// the bit image the contour kernels are handed (after the speck passes), as bytes
extern "C" void proto_speck_clean(const uint8_t* img, int w, int h, uint8_t* out)
{
    const int keep = g_clean;
    g_clean = 1;
    const std::vector<uint32_t> bits = padded_bits(img, w, h);
    g_clean = keep;
    const int wpr = (w + 2 + 31) / 32;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[(size_t)y * w + x] = (bits[(size_t)(y + 1) * wpr + ((x + 1) >> 5)] >> ((x + 1) & 31)) & 1u;
}

// start candidate at padded pixel (px, py): 0 outer, 1 hole (px is the background pixel), -1 none; through the word masks when filtered
static int start_candidate(const BitImage& im, int px, int py, int vx0 = 0, int vx1 = 1 << 20)   // [vx0, vx1]: the columns the caller may look at
{
    if (!g_filter) {
        if (outer_start_candidate(im, px, py)) return 0;
        if (px >= 2 && hole_start_candidate(im, px, py)) return 1;
        return -1;
    }
    const int j = px >> 5;
    const uint32_t* row = im.bits + (size_t)py * im.wpr;
    const uint32_t* up = row - im.wpr;
    const uint32_t cur = row[j], upw = up[j];
    const uint32_t cur_l = (cur << 1) | (j ? row[j - 1] >> 31 : 0u), up_l = (upw << 1) | (j ? up[j - 1] >> 31 : 0u);
    const uint32_t up_r = (upw >> 1) | (j + 1 < im.wpr ? up[j + 1] << 31 : 0u);
    uint32_t mo, mh, valid = 0;
    for (int b = 0; b < 32; b++) if (j * 32 + b >= vx0 && j * 32 + b <= vx1) valid |= 1u << b;
    start_candidate_masks(cur, cur_l, upw, up_l, up_r, true, &mo, &mh, valid);
    if ((mo >> (px & 31)) & 1u) return 0;
    if ((mh >> (px & 31)) & 1u) return 1;
    return -1;
}

extern "C" int proto_find_contours(const uint8_t* img, int w, int h, int32_t* lengths, int max_contours,
                                   int32_t* points, int max_points, int64_t* work_steps)
{
    const int wpr = (w + 2 + 31) / 32;
    const std::vector<uint32_t> bits = padded_bits(img, w, h);
    BitImage im{bits.data(), wpr, w, h};
    struct C { int key; std::vector<uint32_t> pts; };
    std::vector<C> found;
    std::vector<uint32_t> tmp((size_t)w * h * 4 + 16);
    int64_t steps = 0;
    for (int py = 1; py <= h; py++)
        for (int px = 1; px <= w; px++) {
            const int is_hole = start_candidate(im, px, py);
            if (is_hole < 0) continue;
            int n = trace_border(im, px - is_hole, py, is_hole, tmp.data(), (int)tmp.size(), 1 << 30);
            if (n < 0) continue;
            steps += n;
            found.push_back(C{py * 65536 + px, std::vector<uint32_t>(tmp.begin(), tmp.begin() + n)});
        }
    // discovery order = raster order of the transition pixel; output = reverse discovery order
    std::sort(found.begin(), found.end(), [](const C& a, const C& b) { return a.key > b.key; });
    int np = 0;
    for (int i = 0; i < (int)found.size(); i++) {
        if (i < max_contours) lengths[i] = (int)found[i].pts.size();
        for (uint32_t v : found[i].pts) {
            if (np < max_points) { points[2 * np] = (int)(v & 0xffff); points[2 * np + 1] = (int)(v >> 16); }
            np++;
        }
    }
    if (work_steps) *work_steps = steps;
    return (int)found.size();
}


// The relay-segment formulation (aruco_trace.hpp, second half), phase by phase as k_contours_relay runs it:
// grid markers -> small borders from start candidates -> segments -> cyclic lists -> canonical starts -> points.
// stats: [grid markers, max segment length, total segment steps, max segments per cycle, borders, small-phase steps]
extern "C" int proto_find_contours_relay(const uint8_t* img, int w, int h, int kshift, int32_t* lengths,
                                         int max_contours, int32_t* points, int max_points, int64_t* stats)
{
    const int wpr = (w + 2 + 31) / 32, kmask = (1 << kshift) - 1;
    const std::vector<uint32_t> bits = padded_bits(img, w, h);
    BitImage im{bits.data(), wpr, w, h};
    struct C { int key; std::vector<uint32_t> pts; };
    std::vector<C> found;

    // ---- grid markers
    struct M { uint32_t key, cmin; int minoff, next, len; };
    std::vector<M> mk;
    std::unordered_map<uint32_t, int> idx;
    for (int py = 1; py <= h; py++)
        for (int px = 1; px <= w; px++) {
            if (!im.get(px, py)) continue;
            const unsigned ring = ring8(im, px, py);
            if (!ring) continue;
            relay_states_of_pixel(ring, grid_active(ring, px, py, kmask), [&](int s) {
                idx[relay_key(px, py, s)] = (int)mk.size();
                mk.push_back(M{relay_key(px, py, s), 0xffffffffu, 0, -1, 0});
            });
        }

    // ---- small borders: followed whole from their start candidates
    int64_t small_steps = 0;
    for (int py = 1; py <= h; py++)
        for (int px = 1; px <= w; px++) {
            const int is_hole = start_candidate(im, px, py);
            if (is_hole < 0) continue;
            const int sx = px - is_hole, sy = py, start_key = py * 65536 + px;
            RelayWalk wk;
            wk.x = sx; wk.y = sy; wk.n = 0; wk.ring = ring8(im, sx, sy);
            wk.s = relay_start_dir(wk.ring, is_hole);
            if (wk.s < 0) { found.push_back(C{start_key, {relay_point(wk)}}); continue; } // single pixel
            const int s0 = wk.s;
            std::vector<uint32_t> pts;
            bool ok = false;
            for (;;) {
                unsigned run;
                const int d = relay_examine(wk.ring, wk.s, &run);
                small_steps++;
                if (run & grid_active(wk.ring, wk.x, wk.y, kmask)) break;          // a segment walker's border
                if (relay_not_canonical(wk.x, wk.y, run, is_hole, start_key)) break;
                pts.push_back(relay_point(wk));
                relay_advance(im, wk, d);
                if (wk.x == sx && wk.y == sy && wk.s == s0) { ok = true; break; }
            }
            if (ok) found.push_back(C{start_key, pts});
        }

    // ---- segments
    int64_t total = 0, maxseg = 0;
    std::vector<std::vector<uint32_t>> segpts(mk.size());
    for (size_t i = 0; i < mk.size(); i++) {
        M& m = mk[i];
        RelayWalk wk;
        relay_walk_from_key(im, wk, m.key);
        for (;;) {
            unsigned run;
            const int d = relay_examine(wk.ring, wk.s, &run);
            if (wk.n > 0 && (run & grid_active(wk.ring, wk.x, wk.y, kmask))) {
                auto it = idx.find(relay_key(wk.x, wk.y, wk.s));
                if (it == idx.end()) return -1;
                m.next = it->second;
                break;
            }
            if (relay_start_class(wk.ring, run)) {
                const uint32_t k = relay_key(wk.x, wk.y, wk.s);
                if (k < m.cmin) { m.cmin = k; m.minoff = wk.n; }
            }
            segpts[i].push_back(relay_point(wk));
            relay_advance(im, wk, d);
        }
        m.len = wk.n;
        total += m.len;
        maxseg = std::max<int64_t>(maxseg, m.len);
    }

    // ---- cyclic lists: canonical start = smallest start state; rotate the concatenation to it
    std::vector<char> seen(mk.size(), 0);
    int64_t maxcyc = 0;
    for (size_t i0 = 0; i0 < mk.size(); i0++) {
        if (seen[i0]) continue;
        int best = -1, cnt = 0;
        for (int i = (int)i0;;) {
            seen[i] = 1; cnt++;
            if (best < 0 || mk[i].cmin < mk[best].cmin) best = i;
            i = mk[i].next;
            if (i == (int)i0) break;
        }
        maxcyc = std::max<int64_t>(maxcyc, cnt);
        if (mk[best].cmin == 0xffffffffu) return -2; // a border without a start state: the theory would be wrong
        RelayWalk cs;
        relay_walk_from_key(im, cs, mk[best].cmin);
        unsigned run;
        relay_examine(cs.ring, cs.s, &run);
        const int cls = relay_start_class(cs.ring, run);
        if (!cls) return -3;
        C c;
        c.key = cs.y * 65536 + cs.x + (cls == 2 ? 1 : 0);
        std::vector<uint32_t> all;
        for (int i = best;;) {
            all.insert(all.end(), segpts[i].begin(), segpts[i].end());
            i = mk[i].next;
            if (i == best) break;
        }
        const int j = mk[best].minoff, n = (int)all.size();
        c.pts.resize(n);
        for (int o = 0; o < n; o++) c.pts[(o - j + n) % n] = all[o];
        found.push_back(std::move(c));
    }
    std::sort(found.begin(), found.end(), [](const C& a, const C& b) { return a.key > b.key; });
    int np = 0;
    for (int i = 0; i < (int)found.size(); i++) {
        if (i < max_contours) lengths[i] = (int)found[i].pts.size();
        for (uint32_t v : found[i].pts) {
            if (np < max_points) { points[2 * np] = (int)(v & 0xffff); points[2 * np + 1] = (int)(v >> 16); }
            np++;
        }
    }
    if (stats) {
        stats[0] = (int64_t)mk.size(); stats[1] = maxseg; stats[2] = total; stats[3] = maxcyc;
        stats[4] = (int64_t)found.size(); stats[5] = small_steps;
    }
    return (int)found.size();
}


// The TILED relay formulation (aruco_trace.hpp, "TILES"; k_ct_walk / k_ct_lists on the GPU): every tile of whole grid cells finds
// its segments and its small borders from its own pixels plus one pixel on every side -- the rest of the image is filled with
// noise in the tile's copy, so a read outside that window shows up as a wrong result -- and only the cyclic lists are global.
// cells = tile width in grid cells.  stats: [segments, tiles, abandoned segment walks, skipped (neighbour's) segments, borders, abandoned small walks]
// pieces of at most g_cut states per recorded segment (0: whole segments); the kernels use 40 at K = 32
static int g_cut = 0;
extern "C" void proto_set_cut(int cut) { g_cut = cut; }

extern "C" int proto_find_contours_tiled(const uint8_t* img, int w, int h, int kshift, int cells, int32_t* lengths,
                                         int max_contours, int32_t* points, int max_points, int64_t* stats)
{
    const int wpr = (w + 2 + 31) / 32, K = 1 << kshift, kmask = K - 1, cw = cells * K;
    const std::vector<uint32_t> bits = padded_bits(img, w, h);
    struct C { int key; std::vector<uint32_t> pts; };
    std::vector<C> found;
    struct M { uint32_t key, next_key, cmin; int minoff, len, next; std::vector<uint32_t> pts; };
    std::vector<M> mk;
    std::unordered_map<uint32_t, int> idx;
    int64_t n_abandoned = 0, n_skipped = 0, n_small_abandoned = 0, ntiles = 0, n_virtual = 0;
    int rule_err = 0;
    const int nbands = (h + K - 1) / K, ncols = (w + cw - 1) / cw;
    uint32_t lcg = 12345u;
    for (int band = 0; band < nbands; band++)
        for (int col = 0; col < ncols; col++) {
            ntiles++;
            const RelayTile t = relay_tile(w, h, K, cw, band, col);
            // the tile's view: noise outside [x0 - 1, x1 + 1] x [y0 - 1, y1 + 1]
            std::vector<uint32_t> tb(bits.size());
            for (auto& v : tb) { lcg = lcg * 1664525u + 1013904223u; v = lcg; }
            for (int y = std::max(0, t.y0 - 1); y <= std::min(h + 1, t.y1 + 1); y++)
                for (int x = std::max(0, t.x0 - 1); x <= std::min(w + 1, t.x1 + 1); x++) {
                    const uint32_t b = (bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u;
                    uint32_t& v = tb[(size_t)y * wpr + (x >> 5)];
                    v = (v & ~(1u << (x & 31))) | (b << (x & 31));
                }
            BitImage im{tb.data(), wpr, w, h};
            // ---- segments from every marker state of the closed tile
            for (int py = std::max(1, t.y0); py <= std::min(h, t.y1); py++)
                for (int px = std::max(1, t.x0); px <= std::min(w, t.x1); px++) {
                    if (!im.get(px, py)) continue;
                    const unsigned ring = ring8(im, px, py);
                    if (!ring) continue;
                    relay_states_of_pixel(ring, grid_active(ring, px, py, kmask), [&](int s0) {
                        RelayWalk wk;
                        relay_walk_from_key(im, wk, relay_key(px, py, s0));
                        // A walk is recorded in pieces of at most g_cut states (the kernels keep a piece's directions in four registers):
                        // the piece after a cut starts at a state that is no marker, so only the tile that walked up to it knows it --
                        // it is that tile's whatever line it runs on (a neighbour would have had to walk the g_cut states in front of
                        // it, and states two tiles share lie on one grid line: at most K + 1 in a row, g_cut > K + 1).
                        bool after_cut = false;
                        for (;;) {
                            M m{relay_key(wk.x, wk.y, wk.s), 0u, 0xffffffffu, 0, 0, -1, {}};
                            const int n0 = wk.n;
                            bool allbot = !after_cut && wk.y == t.y1, allright = !after_cut && wk.x == t.x1, inside = true, cut_here = false;
                            for (;;) {
                                unsigned run;
                                const int d = relay_examine(wk.ring, wk.s, &run);
                                const int np = wk.n - n0;
                                if (np > 0 && (run & grid_active(wk.ring, wk.x, wk.y, kmask))) { m.next_key = relay_key(wk.x, wk.y, wk.s); break; }
                                if (g_cut > 0 && np == g_cut) { m.next_key = relay_key(wk.x, wk.y, wk.s); cut_here = true; break; }
                                if (relay_start_class(wk.ring, run)) {
                                    const uint32_t k = relay_key(wk.x, wk.y, wk.s);
                                    if (k < m.cmin) { m.cmin = k; m.minoff = np; }
                                }
                                m.pts.push_back(relay_point(wk));
                                const int nx = wk.x + dir_dx(d), ny = wk.y + dir_dy(d);
                                if (!relay_tile_has(t, nx, ny)) { inside = false; break; } // a neighbour's segment
                                relay_advance(im, wk, d);
                                allbot = allbot && wk.y == t.y1; allright = allright && wk.x == t.x1;
                            }
                            if (!inside) { if (after_cut) rule_err = -13; n_abandoned++; return; } // (a walk never leaves its tile after a cut)
                            if (after_cut || relay_tile_owns(t, allbot, allright)) {
                                m.len = wk.n - n0;
                                if (after_cut) n_virtual++;
                                if (idx.count(m.key)) idx[m.key] = -1; // owned twice: reported below
                                else { idx[m.key] = (int)mk.size(); mk.push_back(std::move(m)); }
                            } else {
                                n_skipped++;
                                if (cut_here) rule_err = -14; // (a piece that fills the code is never a neighbour's)
                            }
                            if (!cut_here) return;
                            after_cut = true;
                        }
                    });
                }
            // ---- small borders from the start candidates strictly between the tile's relay rows
            for (int py = std::max(1, t.y0 + 1); py <= std::min(h, t.y1 - 1); py++)
                for (int px = std::max(1, t.x0); px <= std::min(w, t.x1 + 1); px++) { // px: the candidate pixel; the start pixel is px - is_hole
                    int is_hole = start_candidate(im, px, py, t.x0 - 1, t.x1 + 1);
                    if (is_hole == 0 && px > t.x1) is_hole = -1;
                    if (is_hole == 1 && px - 1 < t.x0) is_hole = -1;
                    if (is_hole < 0) continue;
                    const int sx = px - is_hole, sy = py, start_key = py * 65536 + px;
                    RelayWalk wk;
                    wk.x = sx; wk.y = sy; wk.n = 0; wk.ring = ring8(im, sx, sy);
                    wk.s = relay_start_dir(wk.ring, is_hole);
                    if (wk.s < 0) continue; // isolated pixel: below
                    const int s0 = wk.s;
                    std::vector<uint32_t> pts;
                    bool ok = false;
                    for (;;) {
                        unsigned run;
                        const int d = relay_examine(wk.ring, wk.s, &run);
                        if (run & grid_active(wk.ring, wk.x, wk.y, kmask)) break;
                        if (relay_not_canonical(wk.x, wk.y, run, is_hole, start_key)) break;
                        pts.push_back(relay_point(wk));
                        const int nx = wk.x + dir_dx(d), ny = wk.y + dir_dy(d);
                        if (!relay_tile_has(t, nx, ny)) { n_small_abandoned++; break; }
                        relay_advance(im, wk, d);
                        if (wk.x == sx && wk.y == sy && wk.s == s0) { ok = true; break; }
                    }
                    if (ok) found.push_back(C{start_key, pts});
                }
        }
    // isolated pixels have no states: no tile logic applies (and no kernel keeps a one-point border)
    {
        BitImage im{bits.data(), wpr, w, h};
        for (int py = 1; py <= h; py++)
            for (int px = 1; px <= w; px++)
                if (im.get(px, py) && !ring8(im, px, py)) found.push_back(C{py * 65536 + px, {(uint32_t)(px - 1) | ((uint32_t)(py - 1) << 16)}});
    }
    // every marker state of the frame must have exactly one owner
    {
        BitImage im{bits.data(), wpr, w, h};
        size_t nstates = 0;
        int bad = 0;
        for (int py = 1; py <= h; py++)
            for (int px = 1; px <= w; px++) {
                if (!im.get(px, py)) continue;
                const unsigned ring = ring8(im, px, py);
                if (!ring) continue;
                relay_states_of_pixel(ring, grid_active(ring, px, py, kmask), [&](int s) {
                    nstates++;
                    auto it = idx.find(relay_key(px, py, s));
                    if (it == idx.end() || it->second < 0) bad++;
                });
            }
        if (bad) return -10;
        if (rule_err) return rule_err;
        for (auto& kv : idx) if (kv.second < 0) return -15; // a piece after a cut owned twice
        if (nstates + (size_t)n_virtual != mk.size()) return -11;
    }
    for (auto& m : mk) {
        auto it = idx.find(m.next_key);
        if (it == idx.end()) return -1;
        m.next = it->second;
    }
    // no small border twice (a start candidate on a shared column is tried by both tiles)
    {
        std::vector<int> keys;
        for (auto& c : found) keys.push_back(c.key);
        std::sort(keys.begin(), keys.end());
        if (std::adjacent_find(keys.begin(), keys.end()) != keys.end()) return -12;
    }
    // ---- cyclic lists, as in proto_find_contours_relay
    BitImage im{bits.data(), wpr, w, h};
    std::vector<char> seen(mk.size(), 0);
    for (size_t i0 = 0; i0 < mk.size(); i0++) {
        if (seen[i0]) continue;
        int best = -1;
        for (int i = (int)i0;;) {
            seen[i] = 1;
            if (best < 0 || mk[i].cmin < mk[best].cmin) best = i;
            i = mk[i].next;
            if (i == (int)i0) break;
            if (seen[i]) return -4; // not a cycle
        }
        if (mk[best].cmin == 0xffffffffu) return -2;
        RelayWalk cs;
        relay_walk_from_key(im, cs, mk[best].cmin);
        unsigned run;
        relay_examine(cs.ring, cs.s, &run);
        const int cls = relay_start_class(cs.ring, run);
        if (!cls) return -3;
        C c;
        c.key = cs.y * 65536 + cs.x + (cls == 2 ? 1 : 0);
        std::vector<uint32_t> all;
        for (int i = best;;) {
            all.insert(all.end(), mk[i].pts.begin(), mk[i].pts.end());
            i = mk[i].next;
            if (i == best) break;
        }
        const int j = mk[best].minoff, n = (int)all.size();
        c.pts.resize(n);
        for (int o = 0; o < n; o++) c.pts[(o - j + n) % n] = all[o];
        found.push_back(std::move(c));
    }
    std::sort(found.begin(), found.end(), [](const C& a, const C& b) { return a.key > b.key; });
    int np = 0;
    for (int i = 0; i < (int)found.size(); i++) {
        if (i < max_contours) lengths[i] = (int)found[i].pts.size();
        for (uint32_t v : found[i].pts) {
            if (np < max_points) { points[2 * np] = (int)(v & 0xffff); points[2 * np + 1] = (int)(v >> 16); }
            np++;
        }
    }
    if (stats) {
        stats[0] = (int64_t)mk.size(); stats[1] = ntiles; stats[2] = n_abandoned; stats[3] = n_skipped;
        stats[4] = (int64_t)found.size(); stats[5] = n_small_abandoned; stats[6] = n_virtual;
    }
    return (int)found.size();
}
