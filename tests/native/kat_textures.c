/*
 * kat_textures.c -- procedural alpha textures of the reference's known-answer tests.
 *
 * Test infrastructure.  Each generator restates one texture lambda of
 * /root/reference/support/tests/test_omm_bake_cpu.cpp (line cited) so the KAT inputs can be
 * regenerated bit-for-bit at test time (they are too big to commit: 1024x1024 fp32 each).
 * Built by tests/ommtest.py with gcc -O2 -msse4.1 -ffp-contract=off against the image's glibc
 * (sinf / cosf come from the same libm the reference's test binary would use here).
 */
#include <math.h>
#include <stdint.h>

enum {
    KAT_CONST = 0,        /* param = value                                          :796 */
    KAT_CIRCLE = 1,       /* StandardCircle                                         :64-76 */
    KAT_DIAG8 = 2,        /* (i%8 != j%8) ? param : 1-param                         :904-909, :918-923 */
    KAT_CORNER = 3,       /* (0,0) -> 0.6 else 0.4                                  :933-937 */
    KAT_SINE = 4,         /* :1026-1033 */
    KAT_MANDELBROT = 5,   /* :1088-1112 */
    KAT_JULIA = 6,        /* :1214-1241 */
    KAT_UNIFORM4 = 7,     /* :1399-1410 */
    KAT_HEXAGONS = 8,     /* :1432-1444 */
    KAT_CHECKER2 = 9,     /* test_subdiv.cpp:88-93 */
    KAT_SINE_U8 = 10,     /* :1006-1011 (value scaled to [0,255]) */
    KAT_JULIA_U8 = 11     /* :1323-1326 */
};

static float circle(int i, int j, int w)
{
    if (i == 0 && j == 0) return 0.6f;
    const float r = 0.4f;
    const float ux = (float)i / (float)w, uy = (float)j / (float)w;
    const float dx = ux - 0.5f, dy = uy - 0.5f;
    if (sqrtf(dx * dx + dy * dy) < r) return 0.f;
    return 1.f;
}

static float mandelbrot(int i, int j, int w, int h)
{
    const float ux = 1.2f * (float)i / (float)w - 0.1f, uy = 1.2f * (float)j / (float)h - 0.1f;
    const float cx0 = 2.f * ux - 1.f, cy0 = 2.f * uy - 1.f;
    float zx = 0, zy = 0;
    const float cx = cx0 - 0.5f, cy = cy0 - 0.f;
    int inside = 1;
    for (int k = 0; k < 20; k++) {
        const float nx = (zx * zx - zy * zy) + cx, ny = (zx * zy + zy * zx) + cy;
        zx = nx; zy = ny;
        if ((double)sqrtf(zx * zx + zy * zy) > 2.) { inside = 0; break; }
    }
    return inside ? 0.f : 1.f;
}

static float julia(int i, int j, int w, int h)
{
    const float ux = 1.2f * (float)i / (float)w - 0.1f, uy = 1.2f * (float)j / (float)h - 0.1f;
    float z0x = 5.f * (ux - .5f), z0y = 5.f * (uy - .27f);
    float colx = 0.f;
    const float time = 3.1f;
    const float cx = cosf(time) * cosf(time / 2.f), cy = cosf(time) * sinf(time / 2.f);
    for (int k = 0; k < 500; k++) {
        const float zx = (z0x * z0x - z0y * z0y) + cx, zy = (z0x * z0y + z0y * z0x) + cy;
        const float mq = zx * zx + zy * zy;
        if (mq > 4.f) { colx = (float)k / 20.f; break; }
        z0x = zx; z0y = zy;
        colx = mq / 2.f;
    }
    const float cl = colx < 0.f ? 0.f : (1.f < colx ? 1.f : colx); /* std::clamp */
    const float alpha = cl >= 0.5f ? 0.6f : 0.4f;
    return 1.f - alpha;
}

static float hexagons(int i, int j)
{
    const float scale = 30.f, grid = 0.2f;
    float px = scale * (float)i / 1024.f, py = scale * (float)j / 1024.f;
    px *= 0.57735f * 2.0f;
    py += 0.5f * (float)((uint32_t)floorf(px) % 2);
    px = fabsf((px - floorf(px)) - 0.5f);
    py = fabsf((py - floorf(py)) - 0.5f);
    const float a = px * 1.5f + py, b = py * 2.0f;
    const float d = fabsf((a < b ? b : a) - 1.0f);
    float t = (d - 0.0f) / (grid - 0.0f);
    t = t < 0.f ? 0.f : t;   /* glm::clamp = min(max(x, lo), hi) */
    t = 1.f < t ? 1.f : t;
    return t * t * (3.f - 2.f * t);
}

static float eval(int kind, float param, int i, int j, int w, int h)
{
    switch (kind) {
    case KAT_CONST: return param;
    case KAT_CIRCLE: return circle(i, j, w);
    case KAT_DIAG8: return ((i % 8) != (j % 8)) ? param : 1.f - param;
    case KAT_CORNER: return (i == 0 && j == 0) ? 0.6f : 0.4f;
    case KAT_SINE: {
        if (i == 0 && j == 0) return 0.6f;
        const float uv = (float)i / (float)w;
        return 1.f - sinf(uv * 15);
    }
    case KAT_MANDELBROT: return mandelbrot(i, j, w, h);
    case KAT_JULIA: return julia(i, j, w, h);
    case KAT_UNIFORM4: {
        static const float values[4] = { 0.9f, 0.1f, 0.1f, 0.7f };
        return 1.f - values[(i % 2) + 2 * (j % 2)];
    }
    case KAT_HEXAGONS: return hexagons(i, j);
    case KAT_CHECKER2: return ((i % 2) != (j % 2)) ? 0.f : 1.f;
    default: return 0.f;
    }
}

/* out is w*h tightly packed */
void kat_fill_f32(int kind, float param, int w, int h, float* out)
{
    for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i)
            out[i + (long)j * w] = eval(kind, param, i, j, w, h);
}

void kat_fill_u8(int kind, float param, int w, int h, uint8_t* out)
{
    (void)param;
    for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i) {
            uint8_t v = 0;
            if (kind == KAT_SINE_U8) {
                const float uv = (float)i / (float)w;
                const float val = 0.5f - 0.5f * sinf(uv * 15);
                v = (uint8_t)(val * 255.f);
            } else if (kind == KAT_JULIA_U8) {
                const float val = julia(i, j, w, h) * 255.f;
                const float cl = val < 0.f ? 0.f : (255.f < val ? 255.f : val);
                v = (uint8_t)cl;
            }
            out[i + (long)j * w] = v;
        }
}

/* ---- seeded synthetic workload textures (bench.py / scale tests; not reference KATs) ---- */
static uint32_t wl_hash(uint32_t x)
{
    x = (x ^ (x >> 16)) * 0x7feb352du;
    x = (x ^ (x >> 15)) * 0x846ca68bu;
    return x ^ (x >> 16);
}
static float wl_lattice(uint32_t ix, uint32_t iy, uint32_t seed)
{
    return (float)(wl_hash(ix * 73856093u ^ iy * 19349663u ^ seed) >> 8) * (1.0f / 16777216.0f);
}
/* multi-octave value noise in [0,1] */
static float wl_noise(int x, int y, uint32_t seed, int octaves, int baseCell)
{
    float sum = 0.f, amp = 1.f, tot = 0.f;
    for (int o = 0; o < octaves; ++o) {
        int cell = baseCell >> o; if (cell < 1) cell = 1;
        const uint32_t gx = (uint32_t)(x / cell), gy = (uint32_t)(y / cell);
        const float fx = ((float)(x % cell) + 0.5f) / (float)cell, fy = ((float)(y % cell) + 0.5f) / (float)cell;
        const float sx = fx * fx * (3.f - 2.f * fx), sy = fy * fy * (3.f - 2.f * fy);
        const uint32_t s = seed * 83492791u + (uint32_t)o * 2654435761u;
        const float a = wl_lattice(gx, gy, s), b = wl_lattice(gx + 1, gy, s), c = wl_lattice(gx, gy + 1, s), d = wl_lattice(gx + 1, gy + 1, s);
        const float v = (a * (1.f - sx) + b * sx) * (1.f - sy) + (c * (1.f - sx) + d * sx) * sy;
        sum += amp * v; tot += amp; amp *= 0.5f;
    }
    return sum / tot;
}
void wl_noise_f32(uint32_t seed, int w, int h, int octaves, int baseCell, float* out)
{
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) out[x + (long)y * w] = wl_noise(x, y, seed, octaves, baseCell);
}
/* "foliage-style" mask: thresholded low-frequency blobs with a soft edge a few texels wide */
void wl_foliage_u8(uint32_t seed, int w, int h, int feature, uint8_t* out)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float a = (wl_noise(x, y, seed, 3, feature) - 0.5f) * 24.0f + 0.5f;
            a = a < 0.f ? 0.f : (a > 1.f ? 1.f : a);
            out[x + (long)y * w] = (uint8_t)(a * 255.0f + 0.5f);
        }
}
