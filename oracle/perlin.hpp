// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/perlin.rs:5-86 of the reference.
#pragma once
#include <array>
#include <cmath>
#include "rand_gen.hpp"

namespace orc {

struct Perlin {
    static constexpr int POINT_COUNT = 256;
    std::array<V3, POINT_COUNT> rand_float;
    std::array<std::array<int, POINT_COUNT>, 3> perm;

    // perlin.rs:14-22 — gradients are normalised points of the cube [-1,1)^3 (no rejection), then
    // three permutations are generated in order.
    Perlin() {
        for (int i = 0; i < POINT_COUNT; ++i) rand_float[i] = normalize(get_rand_vec3_range(-1.f, 1.f));
        for (int a = 0; a < 3; ++a) perm[a] = perlin_generate_perm();
    }
    // perlin.rs:73-86 — swap target is floor(i * xi) in [0, i-1]: index i never stays in place.
    static std::array<int, POINT_COUNT> perlin_generate_perm() {
        std::array<int, POINT_COUNT> p;
        for (int i = 0; i < POINT_COUNT; ++i) p[i] = i;
        for (int i = POINT_COUNT - 1; i >= 1; --i) {
            size_t target = get_rand_usize_range(0, (size_t)i);
            std::swap(p[i], p[target]);
        }
        return p;
    }
    // perlin.rs:24-43
    float noise(V3 p) const {
        int idx[3] = {(int)std::floor(p.x), (int)std::floor(p.y), (int)std::floor(p.z)};
        V3 c[2][2][2];
        for (int di = 0; di < 2; ++di)
            for (int dj = 0; dj < 2; ++dj)
                for (int dk = 0; dk < 2; ++dk) {
                    int d[3] = {di, dj, dk};
                    int ind = 0;
                    for (int a = 0; a < 3; ++a) ind ^= perm[a][(idx[a] + d[a]) & 0xff];
                    c[di][dj][dk] = rand_float[ind];
                }
        return trilinear_interp(c, p);
    }
    // perlin.rs:44-63 — the Hermite-smoothed fraction is used for the blend weights AND for the
    // offset vector dotted with the gradient.
    static float trilinear_interp(const V3 c[2][2][2], V3 p) {
        float f[3];
        for (int a = 0; a < 3; ++a) {
            float x = p[a] - std::floor(p[a]);
            f[a] = x * x * (3.f - 2.f * x);
        }
        V3 uvw(f[0], f[1], f[2]);
        float accm = 0.f;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    V3 weight = uvw - V3((float)i, (float)j, (float)k);
                    int ijk[3] = {i, j, k};
                    float prod = 1.f;
                    for (int a = 0; a < 3; ++a) {
                        float t = (float)ijk[a];
                        prod = prod * (t * uvw[a] + (1.f - t) * (1.f - uvw[a]));
                    }
                    accm += dot(c[i][j][k], weight) * prod;
                }
        return accm;
    }
    // perlin.rs:65-71
    float turb(V3 p, int depth = 7) const {
        float accum = 0.f, weight = 1.f;
        V3 temp_p = p;
        for (int i = 0; i < depth; ++i) {
            accum = accum + weight * noise(temp_p);
            weight = 0.5f * weight;
            temp_p = 2.f * temp_p;
        }
        return std::fabs(accum);
    }
};

}  // namespace orc
