// aruco_trace.hpp -- border following on a bit-packed binary image, written once for the gfx950 kernels
// (aruco_kernels.hip) and for the host-side unit test of the same logic (tests/proto_contours.cpp).
//
// Reference behaviour being reproduced: cv::findContours(RETR_LIST, CHAIN_APPROX_NONE) as called at
// Thirdparty/aruco/aruco/markerdetector_impl.cpp:3104.  OpenCV's implementation is a SEQUENTIAL raster scan
// (Suzuki-Abe) that marks visited pixels so that every border is followed exactly once, from its raster-first
// pixel.  Here every border is followed independently and read-only:
//
//   * a pixel can only be the raster-first pixel of an outer border if it is foreground, its W, NW, N, NE
//     neighbours are background ("local top");
//   * a background pixel can only be the raster-first pixel of a hole if its W and N neighbours are foreground;
//   * such a candidate is the real start iff following its border never meets a raster-smaller foreground pixel
//     (outer) / never examines a raster-smaller background pixel (hole); otherwise the walk is abandoned.
//
// The walk itself is icvFetchContour (OpenCV 3.4 contours.cpp) minus the marking.
#pragma once
#include <stdint.h>

#include "../../include/orbfe_math.h"

namespace orbfe {

// row * words-per-row on the device as a 24-bit multiply: it folds with the word index into one v_mad_u32_u24 (the 32-bit product
// the compiler emits for `y * wpr` -- it cannot bound the operands -- needs a separate addition); ring8() runs once per walk step
#if defined(__HIP_DEVICE_COMPILE__)
#define ORBFE_ROWMUL(y, wpr) __mul24((y), (wpr))
#else
#define ORBFE_ROWMUL(y, wpr) ((y) * (wpr))
#endif

// Bit image with a one-pixel zero frame: pixel (x, y) of the W x H image is bit (x+1) of row (y+1).
struct BitImage {
    const uint32_t* bits;
    int wpr; // 32-bit words per row
    int W, H;
    ORBFE_HD int get(int px, int py) const // padded coordinates, 0 <= px < W+2, 0 <= py < H+2
    {
        return (bits[ORBFE_ROWMUL(py, wpr) + (px >> 5)] >> (px & 31)) & 1;
    }
};

// direction codes 0..7 = E, NE, N, NW, W, SW, S, SE (OpenCV icvCodeDeltas)
// dx + 1 / dy + 1 of the 8 directions packed two bits each (dx: 1 1 0 -1 -1 -1 0 1; dy: 0 -1 -1 -1 0 1 1 1)
ORBFE_HD int dir_dx(int s) { return (int)((((2u) | (2u << 2) | (1u << 4) | (0u << 6) | (0u << 8) | (0u << 10) | (1u << 12) | (2u << 14)) >> (2 * s)) & 3u) - 1; }
ORBFE_HD int dir_dy(int s) { return (int)((((1u) | (0u << 2) | (0u << 4) | (0u << 6) | (1u << 8) | (2u << 10) | (2u << 12) | (2u << 14)) >> (2 * s)) & 3u) - 1; }

// The 8 neighbours of padded pixel (x, y) as one byte, bit k = direction k (E, NE, N, NW, W, SW, S, SE).
// Three 64-bit funnel reads; the bit array must have one spare word after the last row.
ORBFE_HD unsigned ring8(const BitImage& im, int x, int y)
{
    const int sh = (x - 1) & 31, w0 = (x - 1) >> 5;
    const uint32_t* r0 = im.bits + (ORBFE_ROWMUL(y - 1, im.wpr) + w0);
    const uint32_t* r1 = r0 + im.wpr;
    const uint32_t* r2 = r1 + im.wpr;
    const unsigned a = (unsigned)((((unsigned long long)r0[1] << 32) | r0[0]) >> sh) & 7u; // row y-1: x-1, x, x+1
    const unsigned b = (unsigned)((((unsigned long long)r1[1] << 32) | r1[0]) >> sh) & 7u; // row y
    const unsigned c = (unsigned)((((unsigned long long)r2[1] << 32) | r2[0]) >> sh) & 7u; // row y+1
    // E = b2; NE, N, NW = a2, a1, a0 (= bit-reversed a); W = b0; SW, S, SE = c0, c1, c2
    const unsigned rev3 = (0x00ee9ca0u >> (3 * a)) & 7u; // 3-bit reversal table: 0 4 2 6 1 5 3 7
    return (b >> 2) | (rev3 << 1) | ((b & 1u) << 4) | (c << 5);
}

ORBFE_HD int ctz8(unsigned v) // v != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)v) - 1;
#else
    return __builtin_ctz(v);
#endif
}

// Border following as an explicit state machine, so that a GPU lane can advance its walk by ONE step per loop
// iteration and pick up a new candidate as soon as its walk ends (no lane waits for the longest walk of its wave).
struct TraceState {
    int sx, sy, is_hole, start_key;
    int i1x, i1y; // first neighbour found from the start pixel (termination test)
    int i3x, i3y; // current pixel
    int s;        // direction from which the current pixel was entered (+4), OpenCV's `s`
    int n;        // points emitted so far
    unsigned ring;
};

// (sx, sy): start pixel in PADDED coordinates; is_hole selects the hole start rule; for holes (sx+1, sy) is the
// hole's candidate raster-first background pixel.  Returns 1 if the border is a single pixel (n = 1, done), else 0.
ORBFE_HD int trace_init(const BitImage& im, TraceState& t, int sx, int sy, int is_hole)
{
    t.sx = sx; t.sy = sy; t.is_hole = is_hole;
    t.start_key = is_hole ? (sy * 65536 + sx + 1) : (sy * 65536 + sx);
    int s_end, s;
    s_end = s = is_hole ? 0 : 4;
    t.ring = ring8(im, sx, sy);
    do {
        s = (s - 1) & 7;
    } while (((t.ring >> s) & 1u) == 0 && s != s_end);
    t.n = 0;
    t.i3x = sx; t.i3y = sy;
    if (s == s_end) { t.n = 1; return 1; }
    t.i1x = sx + dir_dx(s); t.i1y = sy + dir_dy(s);
    t.s = s;
    return 0;
}

// One step.  *pt receives the emitted point (x | y << 16, image coordinates).  Returns 0 = continue,
// 1 = border closed (t.n = length), -1 = met a raster-smaller pixel (not the canonical start).
// "Search counter-clockwise from the direction after the previous pixel for the first foreground neighbour" is a
// rotate + count-trailing-zeros on the 3x3 neighbourhood byte.
ORBFE_HD int trace_step(const BitImage& im, TraceState& t, uint32_t* pt)
{
    const int rot = t.s + 1; // 1..8: first direction examined
    const unsigned m = t.ring | (t.ring << 8) | (t.ring << 16);
    const unsigned r = (m >> rot) & 0xffu; // bit i = direction (rot + i) & 7; the previous pixel guarantees r != 0
    const int i = ctz8(r);
    const int s = (rot + i) & 7;
    const int key3 = t.i3y * 65536 + t.i3x;
    if (t.is_hole) {
        // examined background 4-neighbours belong to the hole being followed
        const unsigned exr = (1u << i) - 1u;
        const unsigned ex = ((exr << rot) | ((exr << rot) >> 8)) & 0xffu;
        if (((ex & 0x04u) && key3 - 65536 < t.start_key) || // N
            ((ex & 0x10u) && key3 - 1 < t.start_key) ||     // W
            ((ex & 0x01u) && key3 + 1 < t.start_key) ||     // E
            ((ex & 0x40u) && key3 + 65536 < t.start_key))   // S
            return -1;
    } else if (key3 < t.start_key) {
        return -1;
    }
    *pt = (uint32_t)(t.i3x - 1) | ((uint32_t)(t.i3y - 1) << 16);
    t.n++;
    const int i4x = t.i3x + dir_dx(s), i4y = t.i3y + dir_dy(s);
    if (i4x == t.sx && i4y == t.sy && t.i3x == t.i1x && t.i3y == t.i1y) return 1;
    t.i3x = i4x;
    t.i3y = i4y;
    t.ring = ring8(im, i4x, i4y);
    t.s = (s + 4) & 7;
    return 0;
}

// Whole-border convenience wrapper (host prototype, and the point-emitting second pass on the GPU).
// out (may be null) receives at most out_cap points.  Returns the length, -1 (not canonical) or -2 (max_steps).
ORBFE_HD int trace_border(const BitImage& im, int sx, int sy, int is_hole, uint32_t* out, int out_cap, int max_steps)
{
    TraceState t;
    if (trace_init(im, t, sx, sy, is_hole)) {
        if (out && out_cap > 0) out[0] = (uint32_t)(sx - 1) | ((uint32_t)(sy - 1) << 16);
        return 1;
    }
    for (;;) {
        uint32_t pt;
        const int n0 = t.n;
        const int st = trace_step(im, t, &pt);
        if (st < 0) return -1;
        if (out && n0 < out_cap) out[n0] = pt;
        if (st == 1) return t.n;
        if (t.n > max_steps) return -2;
    }
}

// Candidate tests on the padded bit image (px, py in 1..W, 1..H).
ORBFE_HD bool outer_start_candidate(const BitImage& im, int px, int py)
{
    return im.get(px, py) && !im.get(px - 1, py) && !im.get(px - 1, py - 1) && !im.get(px, py - 1) &&
           !im.get(px + 1, py - 1);
}
// (px, py) is the BACKGROUND pixel; the border start is (px-1, py)
ORBFE_HD bool hole_start_candidate(const BitImage& im, int px, int py)
{
    return !im.get(px, py) && im.get(px - 1, py) && im.get(px, py - 1);
}

// ------------------------------------------------------------------------------------------------------------------
// FEWER WALKS.  The detector keeps only borders of more than 70 points (markerdetector_impl.cpp:3046 / :3217), and on textured
// frames nearly every start candidate belongs to a border that is dropped: specks of the thresholded texture and the ragged edges of
// large regions (a 640 x 480 stream frame: 12.8 k candidates, 8.0 k of its 8.2 k components have at most 17 pixels).  Two exact
// reductions, both decided on whole 32-pixel words:
//
// (1) RUN TESTS on the start candidates.  A candidate is the canonical start of a border only if its pixel is the raster-first pixel
//     of its component (outer) / of its background region (hole).  Follow the row eastwards from the candidate along its run of
//     foreground (background) pixels: a foreground pixel among the NW, N, NE neighbours of the run belongs to the same 8-connected
//     component, a background pixel N of the run to the same 4-connected region -- both raster-smaller, so the candidate cannot be
//     canonical and no walk is started.  Runs are cut at the word's end (a carry that leaves the word proves nothing: the walk decides).
//     "The carry of start bit p through the run's good pixels stops on a bad pixel" is one addition; getting from the stop back to p
//     is the same addition on the bit-reversed word.
//
// (2) SPECKS.  A border visits a pixel at most four times -- the visits of one border to a pixel p are states (p, s) with pairwise
//     different runs, a run on a followed border is never empty (an empty run's predecessor has an empty run, or the cycle is the
//     three-pixel loop inside a solid corner: tests/test_contour_logic_cpu.py), and the 8 neighbours of p hold at most four maximal
//     background runs -- so a component of n pixels has no border, outer or hole, longer than 4 n points.  A w x h window whose
//     one-pixel rim is empty isolates what is inside it from the rest of the frame; with w h <= 17 nothing inside can have a border
//     of more than 68 points, and clearing the window changes no other border (a walk only ever examines its own component and the
//     background region it follows).  speck_* below are one PASS for one window shape, on the padded bit image:
//         anchor (ax, ay), ax >= 0: the rim's top-left corner; rim = columns ax and ax + w + 1, rows ay and ay + h + 1;
//         empty rim -> pixels (ax + 1 .. ax + w, ay + 1 .. ay + h) are cleared;
//     the image is taken as zero beyond its frame.  Passes are applied one after the other (what one clears opens rims for the next);
//     a pixel of a pass's output depends on the input within w columns and h rows of it.
ORBFE_HD uint32_t bitrev32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
#endif
}

// start bits of `starts` (a subset of `good`) whose run of consecutive `good` bits towards bit 31 ends on a bit of `bad`
ORBFE_HD uint32_t run_ends_on(uint32_t starts, uint32_t good, uint32_t bad)
{
    const uint32_t land = (good + starts) & ~good & bad;                 // where a start's carry stopped, if that is a bad pixel
    const uint32_t rg = bitrev32(good);
    const uint32_t back = (rg + (bitrev32(land) << 1)) & ~rg;            // the carry back down the run stops one below its start
    return bitrev32(back) << 1;
}

// Start candidates of one word of row y of the padded image (bit i = pixel 32 j + i): cur = the row's word, cur_l = the W neighbours of
// its pixels, upw / up_l / up_r = their N / NW / NE neighbours.  *outer: pixels that may start an outer border; *hole: BACKGROUND pixels b
// whose W neighbour may start a hole border (the start pixel is b - 1).  filter: apply the run tests (1) above.
// valid: the pixels of the word the caller may look at (a tile: its window; the kernels hold whole words: all ones) -- a run is cut there too.
ORBFE_HD void start_candidate_masks(uint32_t cur, uint32_t cur_l, uint32_t upw, uint32_t up_l, uint32_t up_r, bool filter, uint32_t* outer,
                                    uint32_t* hole, uint32_t valid = 0xffffffffu)
{
    uint32_t mo = cur & ~cur_l & ~up_l & ~upw & ~up_r;
    uint32_t mh = ~cur & cur_l & upw;
    if (filter) {
        // foreground with foreground above: not the first of its component.  (The NE neighbour of the word's last pixel lies in another
        // word of the row above, which not every caller holds for every word it looks at: left out, so that all callers decide alike.)
        const uint32_t bad_o = cur & (up_l | upw | (up_r & 0x7fffffffu)) & valid;
        mo &= ~run_ends_on(mo, cur & ~bad_o & valid, bad_o);
        const uint32_t bad_h = ~cur & ~upw & valid;                     // background with background above
        mh &= ~run_ends_on(mh, ~cur & upw & valid, bad_h);
    }
    *outer = mo;
    *hole = mh;
}

// ((hi : lo) >> k) & 0xffffffff, 0 <= k <= 31
ORBFE_HD uint32_t funnel_shr(uint32_t hi, uint32_t lo, int k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)k);
#else
    return (uint32_t)((((unsigned long long)hi << 32) | lo) >> k);
#endif
}
// One row's contribution to the rims of the anchors in its word: bit a of *full = one of pixels a .. a + WW + 1 of the row is set (a rim's
// top or bottom row), bit a of *side = pixel a or pixel a + WW + 1 is set (its two columns).  p = the word, pn = the next word of the row.
template <int WW>
ORBFE_HD void speck_row_masks(uint32_t p, uint32_t pn, uint32_t* full, uint32_t* side)
{
    uint32_t f = p, last = p;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 1; k <= WW + 1; k++) { last = funnel_shr(pn, p, k); f |= last; }
    *full = f;
    *side = p | last;
}
// pixels cleared in a word of a row: e = the anchors (empty rims) of the HH rows above it in this word, el = in the word before it
template <int WW>
ORBFE_HD uint32_t speck_dilate(uint32_t e, uint32_t el)
{
    uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int dx = 1; dx <= WW; dx++) c |= funnel_shr(e, el, 32 - dx);
    return c;
}
// the window shapes of the two passes (w x h <= 17 each); vertical reach of both together
#define ORBFE_SPECK_W1 3
#define ORBFE_SPECK_H1 5
#define ORBFE_SPECK_W2 5
#define ORBFE_SPECK_H2 3
#define ORBFE_SPECK_REACH (ORBFE_SPECK_H1 + ORBFE_SPECK_H2)

// ------------------------------------------------------------------------------------------------------------------
// Relay segments: the same borders, cut into short independent pieces (k_contours_relay, tests/proto_contours.cpp).
//
// A walk state is (pixel p, direction s of the previous border pixel).  One step searches counter-clockwise from s+1
// for the first foreground neighbour; the background directions it examines before finding it are the state's RUN.
// The step function is a bijection on states, so every state lies on exactly one cycle; a cycle that contains a state
// whose run holds a 4-neighbour direction is a border Suzuki-Abe can follow (the border between the pixel's
// component and the background region of that 4-neighbour).
//
// GRID MARKERS.  A state is a grid marker if its run contains W or E and its pixel lies on a relay row
// (y % K == 0), or contains N or S and its pixel lies on a relay column (x % K == 0).  Grid markers are recognisable
// from the 3x3 neighbourhood alone, both when enumerating them up front and when a walk arrives at one.  One lane walks
// from its marker to the next marker (a SEGMENT, about K steps on straight edges); the segments of a border form a
// cyclic list, and no lane ever walks a long border alone.
//
// CANONICAL START.  OpenCV's raster scan starts an outer border at the component's raster-first pixel (a "local top":
// W, NW, N, NE background; the state whose run contains W) and a hole border at W(b), b the raster-first pixel of
// the hole (E background, NE foreground; the state whose run contains E).  Call states of these two patterns START
// states.  On every border cycle the canonical start is the start state with the smallest state key:
//   * outer border: the component's raster-first pixel is a local top; hole-pattern start states on the cycle sit on
//     other pixels of the component, hence are larger;
//   * hole border of background region R with raster-first pixel b: a local top c on the cycle has N(c) in R, so
//     b < N(c) and W(b) < c; another hole-pattern start W(b'), b' in R, has b' > b.
// So each segment records the smallest start state it passes and where; the minimum over the cyclic list names the
// canonical start, and its pattern says whether the border is an outer border or a hole border.
//
// Borders that touch no grid marker are small (they live between grid lines); they are followed whole from their
// start candidates with the abandon rule of the first half of this file, and a walk that meets a grid marker stops
// (that border belongs to the segment walkers).

// Grid-marker directions (bit d = direction d; only background directions survive) of padded pixel (x, y).
ORBFE_HD unsigned grid_active(unsigned ring, int x, int y, int kmask)
{
    unsigned a = 0;
    if ((y & kmask) == 0) a |= 0x11u;
    if ((x & kmask) == 0) a |= 0x44u;
    return a & ~ring;
}

// One step's search from state (ring, s): direction of the next border pixel and the run of examined directions.
ORBFE_HD int relay_examine(unsigned ring, int s, unsigned* run)
{
    const int rot = s + 1;
    const unsigned m = ring | (ring << 8) | (ring << 16);
    const unsigned r = (m >> rot) & 0xffu;
    const int i = ctz8(r);
    const unsigned exr = (1u << i) - 1u;
    *run = ((exr << rot) | ((exr << rot) >> 8)) & 0xffu;
    return (rot + i) & 7;
}

// Start pattern of a state with the given run: 1 = outer-border start, 2 = hole-border start, 0 = neither.
ORBFE_HD int relay_start_class(unsigned ring, unsigned run)
{
    if ((run & 0x10u) && (ring & 0x1eu) == 0) return 1;
    if ((run & 0x01u) && (ring & 0x02u)) return 2;
    return 0;
}

// state key: y | x | s (padded coordinates), ordered like the raster key of the pixel; never 0
ORBFE_HD uint32_t relay_key(int x, int y, int s) { return ((uint32_t)y << 16) | ((uint32_t)x << 3) | (uint32_t)s; }

// Calls emit(s) for every state of a pixel whose run contains one of the directions in `a` (a subset of the pixel's
// background directions).  ring != 0: isolated pixels have no states.
template <class F>
ORBFE_HD void relay_states_of_pixel(unsigned ring, unsigned a, F emit)
{
    while (a) {
        const int d = ctz8(a);
        const unsigned rr = ((ring | (ring << 8)) >> d) & 0xffu; // bit j = direction d + j; bit 0 is background
        const int j = 31 - __builtin_clz(rr);                     // highest set bit (ring != 0)
        const int s = (d + j) & 7; // first foreground direction clockwise from d
        unsigned run;
        relay_examine(ring, s, &run);
        a &= ~run;
        emit(s);
    }
}

struct RelayWalk {
    int x, y, s; // padded pixel, direction of the previous border pixel
    int n;       // points emitted so far
    unsigned ring;
};

ORBFE_HD void relay_walk_from_key(const BitImage& im, RelayWalk& w, uint32_t key)
{
    w.x = (int)((key >> 3) & 0x1fffu); w.y = (int)(key >> 16); w.s = (int)(key & 7u);
    w.n = 0;
    w.ring = ring8(im, w.x, w.y);
}

ORBFE_HD uint32_t relay_point(const RelayWalk& w) { return (uint32_t)(w.x - 1) | ((uint32_t)(w.y - 1) << 16); }

ORBFE_HD void relay_advance(const BitImage& im, RelayWalk& w, int d)
{
    w.x += dir_dx(d); w.y += dir_dy(d);
    w.s = (d + 4) & 7;
    w.n++;
    w.ring = ring8(im, w.x, w.y);
}

// The abandon rule of trace_step() in terms of a state's run: true if the walk from a start candidate with raster key
// start_key proves that the candidate is not the canonical start.
ORBFE_HD bool relay_not_canonical(int x, int y, unsigned run, int is_hole, int start_key)
{
    const int key3 = y * 65536 + x;
    if (is_hole)
        return ((run & 0x04u) && key3 - 65536 < start_key) || ((run & 0x10u) && key3 - 1 < start_key) ||
               ((run & 0x01u) && key3 + 1 < start_key) || ((run & 0x40u) && key3 + 65536 < start_key);
    return key3 < start_key;
}

// First state of a start candidate (the state trace_init() sets up): pixel (sx, sy), previous = first foreground
// neighbour clockwise from W (outer) / from E (hole).  Returns -1 for an isolated pixel.
ORBFE_HD int relay_start_dir(unsigned ring, int is_hole)
{
    if (!ring) return -1;
    const int d0 = is_hole ? 0 : 4;
    const unsigned rr = ((ring | (ring << 8)) >> d0) & 0xffu;
    const int j = 31 - __builtin_clz(rr); // highest set bit; direction d0 itself (bit 0) is background for a start candidate
    return (d0 + j) & 7;
}

// ------------------------------------------------------------------------------------------------------------------
// TILES: the relay formulation cut into independent pieces of the frame (k_ct_walk, tests/proto_contours.cpp).
//
// A segment -- the states from one grid marker up to the next -- never crosses a relay row or column: passing from one side of
// a relay row to the other turns counter-clockwise through W or E on a pixel of that row, which makes that state a marker (the
// same with N / S on a relay column).  So the pixels of a segment, its end marker included, lie in one CLOSED grid cell (the
// cell with its four grid lines), and the same holds for a border that touches no marker at all.  A tile is a closed rectangle
// of whole cells, [x0, x1] x [y0, y1] in padded coordinates with all four numbers multiples of K; a workgroup that holds the
// tile and one more pixel on every side can therefore do, without looking at anything else,
//   * every segment whose pixels lie in the tile, and
//   * every small border whose pixels lie in the tile.
// A walk that leaves the tile is abandoned: it is a neighbour's.  Pixels on the grid line between two tiles belong to both, so
// a segment (at least two pixels) that lies entirely on a shared line is seen by both: it goes to the tile below / to the right
// of the line (the one for which the line is y0 / x0), i.e. a tile skips what lies entirely on its bottom row or right column
// when a neighbour exists there.  A small border cannot lie entirely on a grid line (its ends would be markers); its start
// candidate may lie on a shared COLUMN, then both tiles try and the one on whose side the border runs completes it.  Start
// candidates on relay ROWS are never small borders: the start state's run holds W (outer) or E (hole).
struct RelayTile {
    int x0, y0, x1, y1; // closed rectangle, padded coordinates
    int right, lower;   // a neighbour tile exists beyond x1 / beyond y1
};
// tile (band, col) of a W x H image: bands of K rows, cw columns per tile (a multiple of K); ceil(H / K) x ceil(W / cw) tiles
ORBFE_HD RelayTile relay_tile(int W, int H, int K, int cw, int band, int col)
{
    RelayTile t;
    t.x0 = col * cw; t.y0 = band * K; t.x1 = t.x0 + cw; t.y1 = t.y0 + K;
    t.right = t.x1 < W ? 1 : 0;
    t.lower = t.y1 < H ? 1 : 0;
    return t;
}
ORBFE_HD bool relay_tile_has(const RelayTile& t, int x, int y) { return x >= t.x0 && x <= t.x1 && y >= t.y0 && y <= t.y1; }
// a finished segment all of whose pixels had y == y1 (allbot) / x == x1 (allright): does this tile keep it?
ORBFE_HD bool relay_tile_owns(const RelayTile& t, bool allbot, bool allright) { return !((t.lower && allbot) || (t.right && allright)); }

} // namespace orbfe
