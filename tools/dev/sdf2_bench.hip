// Development harness for the second-generation fused SDF kernel (csrc/k_sdf_fwd2.h): builds the chunk streams on the
// host from random weights, checks the kernel against a double-precision host evaluation of the same network
// (reference dpt_models/fields.py:72-108 semantics: Softplus(beta=100), skip at layer 4, analytic input gradient) and
// times v1 (k_sdf_fwd.h) and v2 variants interleaved in ONE process (HIP events on the launch stream).
// Not part of the product library.   usage: sdf2_bench [points=65536] [rounds=10]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>
#include "../../include/vdn_render.h"
// v2 variants are compiled as separate objects (sdf2_variant.hip, one per -D set) and v1 comes from libvdn_render.so
#define V2(TAG, M, S) extern "C" int sdf2_launch_##TAG(const VdnSdfArgs*, hipStream_t);
#include "sdf2_variants.inc"
#undef V2

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

struct Mat { int rows, cols; std::vector<float> w, b; };

// one chunk-stream layer: padded k -> source column (kmap), padded row -> source row (nmap); transposed swaps the roles
struct LayerSpec { const Mat* m; std::vector<int> kmap, nmap; float scale, bias_scale; bool bias, transposed; };

static std::vector<int> ident(int n, int pad) { std::vector<int> v(pad, -1); for (int i = 0; i < n; ++i) v[i] = i; return v; }

static void append_layer(std::vector<char>& blob, const LayerSpec& L, int stride, const float* tail = nullptr) {
    const int kt = (int)L.kmap.size() / 32;
    for (size_t n0 = 0; n0 < L.nmap.size(); n0 += 32) {
        const size_t off = blob.size();
        blob.resize(off + stride, 0);
        unsigned short* out = reinterpret_cast<unsigned short*>(blob.data() + off);
        auto val = [&](int i, int k) -> float {
            const int r = L.nmap[n0 + i], c = L.kmap[k];
            if (r < 0 || c < 0) return 0.0f;
            return L.scale * (L.transposed ? L.m->w[(size_t)c * L.m->cols + r] : L.m->w[(size_t)r * L.m->cols + c]);
        };
        for (int s = 0; s < kt * 2; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                for (int j = 0; j < 8; ++j) out[(s * 64 + lane) * 8 + j] = f2bf(val(i, 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)));
            }
        float* b = reinterpret_cast<float*>(blob.data() + off + kt * 2048);
        for (int i = 0; i < 32; ++i) {
            const int r = L.nmap[n0 + i];
            b[i] = (L.bias && r >= 0) ? L.bias_scale * L.m->b[r] : 0.0f;
        }
        if (tail != nullptr) memcpy(blob.data() + off + 9 * 2048 + 1024, tail, 256 * 4);      // row 0 of W8 rides in every chunk's tail
    }
}

// the 16x16x32 image of a chunk stream (k_sdf_fwd2.h, shape 1): a permutation of each chunk's 16-byte fragment units -
// new fragment 2T + fh, lane r16 + 16 q  <-  old fragment 2T + (q >> 1), lane phi(fh, r16) + 32 (q & 1),
// phi(fh, 4 q' + i) = 16 (q' >> 1) + 8 fh + 4 (q' & 1) + i; bias block and tail unchanged
static std::vector<char> to_s16(const std::vector<char>& blob, int stride, const std::vector<int>& kts) {
    std::vector<char> out = blob;
    for (size_t ch = 0; ch < kts.size(); ++ch) {
        const char* src = blob.data() + ch * stride;
        char* dst = out.data() + ch * stride;
        for (int T = 0; T < kts[ch]; ++T)
            for (int fh = 0; fh < 2; ++fh)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r16 = lane & 15, q = lane >> 4, qq = r16 >> 2, i = r16 & 3;
                    const int phi = 16 * (qq >> 1) + 8 * fh + 4 * (qq & 1) + i;
                    memcpy(dst + ((2 * T + fh) * 64 + lane) * 16, src + ((2 * T + (q >> 1)) * 64 + phi + 32 * (q & 1)) * 16, 16);
                }
    }
    return out;
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 65536;
    const int rounds = argc > 2 ? atoi(argv[2]) : 10;
    const int d0 = 39;
    const float C1 = 144.26950408889634f;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    // weights shaped like the shipped SDF network (fields.py:10-70): dims [39, 256 x 8, 257], layer 3 has 217 outputs
    Mat lin[9];
    const int din[9] = {39, 256, 256, 256, 256, 256, 256, 256, 256};
    const int dout[9] = {256, 256, 256, 217, 256, 256, 256, 256, 257};
    for (int l = 0; l < 9; ++l) {
        lin[l].rows = dout[l]; lin[l].cols = din[l];
        lin[l].w.resize((size_t)dout[l] * din[l]); lin[l].b.resize(dout[l]);
        const float sd = std::sqrt(2.0f / dout[l]);
        for (auto& x : lin[l].w) x = nd(rng) * sd * (l == 0 ? 0.6f : 1.0f);
        for (auto& x : lin[l].b) x = nd(rng) * 0.02f;
        if (l == 8) { for (int c = 0; c < 256; ++c) lin[8].w[c] = std::sqrt(3.14159f) / 16.0f + nd(rng) * 1e-3f; lin[8].b[0] = -0.5f; }
    }
    // forward layer maps (vdn_hip/images.py: sdf_streams)
    struct FW { std::vector<int> km, nm; float sc; };
    FW fw[8];
    for (int l = 0; l < 8; ++l) {
        if (l == 0) fw[l] = {ident(d0, 64), ident(256, 256), 1.0f};
        else if (l == 3) fw[l] = {ident(256, 256), ident(217, 224), 1.0f};
        else if (l == 4) {
            std::vector<int> km(288, -1);
            for (int i = 0; i < 217; ++i) km[i] = i;
            for (int i = 0; i < d0; ++i) km[224 + i] = 217 + i;
            fw[l] = {km, ident(256, 256), (float)(1.0 / std::sqrt(2.0))};
        } else fw[l] = {ident(256, 256), ident(256, 256), 1.0f};
    }
    std::vector<int> nm8(288, -1);
    for (int i = 0; i < 256; ++i) nm8[i] = 1 + i;
    nm8[256] = 0;
    const int stride = 20480;
    // v1 streams ('sdf', 'full') and v2 streams ('sdf2', 'full2')
    std::vector<char> b_sdf, b_full, b_sdf2, b_full2;
    for (int l = 0; l < 8; ++l) {
        LayerSpec L{&lin[l], fw[l].km, fw[l].nm, fw[l].sc, 1.0f, true, false};
        append_layer(b_sdf, L, stride); append_layer(b_full, L, stride);
        LayerSpec L2 = L; L2.bias_scale = C1;
        // residue slots: the 25 spare contraction slots behind the 39 encoded inputs read the first 25 columns again
        if (l == 0) for (int i = 0; i < 25; ++i) L2.kmap[39 + i] = i;
        if (l == 4) for (int i = 0; i < 25; ++i) L2.kmap[224 + 39 + i] = 217 + i;
        append_layer(b_sdf2, L2, stride, lin[8].w.data()); append_layer(b_full2, L2, stride, lin[8].w.data());
    }
    append_layer(b_sdf, LayerSpec{&lin[8], ident(256, 256), ident(1, 32), 1.0f, 1.0f, true, false}, stride);
    append_layer(b_full, LayerSpec{&lin[8], ident(256, 256), nm8, 1.0f, 1.0f, true, false}, stride);
    append_layer(b_sdf2, LayerSpec{&lin[8], ident(256, 256), ident(1, 32), 1.0f / C1, 1.0f, true, false}, stride, lin[8].w.data());
    append_layer(b_full2, LayerSpec{&lin[8], ident(256, 256), nm8, 1.0f / C1, 1.0f, true, false}, stride, lin[8].w.data());
    for (int l = 7; l >= 0; --l) {
        LayerSpec L{&lin[l], fw[l].nm, fw[l].km, fw[l].sc, 1.0f, false, true};   // contraction over the forward OUTPUT order
        append_layer(b_full, L, stride);
        LayerSpec L2 = L; L2.scale = fw[l].sc / 255.0f;                              // v2 sweeps 255 sigma
        append_layer(b_full2, L2, stride, lin[8].w.data());
    }
    // k-tiles per chunk of the v2 streams, in stream order
    std::vector<int> kt_sdf2, kt_full2;
    {
        const int fk[9] = {2, 8, 8, 8, 9, 8, 8, 8, 8}, fn[8] = {8, 8, 8, 7, 8, 8, 8, 8};
        for (int l = 0; l < 8; ++l) for (int t = 0; t < fn[l]; ++t) { kt_sdf2.push_back(fk[l]); kt_full2.push_back(fk[l]); }
        kt_sdf2.push_back(8);
        for (int t = 0; t < 9; ++t) kt_full2.push_back(8);
        for (int l = 7; l >= 0; --l) for (int t = 0; t < fk[l]; ++t) kt_full2.push_back(fn[l]);
    }
    if (kt_sdf2.size() * stride != b_sdf2.size() || kt_full2.size() * stride != b_full2.size()) { printf("chunk table mismatch\n"); return 1; }
    const std::vector<char> b_sdf2x = to_s16(b_sdf2, stride, kt_sdf2), b_full2x = to_s16(b_full2, stride, kt_full2);
    printf("streams: sdf %zu full %zu sdf2 %zu full2 %zu chunks\n", b_sdf.size() / stride, b_full.size() / stride, b_sdf2.size() / stride, b_full2.size() / stride);

    // inputs: points in the unit ball region
    std::vector<float> pts((size_t)P * 3);
    std::uniform_real_distribution<float> ud(-1.0f, 1.0f);
    for (auto& x : pts) x = ud(rng);

    auto dev = [&](const void* src, size_t bytes) { void* d; CK(hipMalloc(&d, bytes)); if (src) CK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice)); else CK(hipMemset(d, 0, bytes)); return d; };
    char* d_sdf = (char*)dev(b_sdf.data(), b_sdf.size());
    char* d_full = (char*)dev(b_full.data(), b_full.size());
    char* d_sdf2 = (char*)dev(b_sdf2.data(), b_sdf2.size());
    char* d_full2 = (char*)dev(b_full2.data(), b_full2.size());
    char* d_sdf2x = (char*)dev(b_sdf2x.data(), b_sdf2x.size());
    char* d_full2x = (char*)dev(b_full2x.data(), b_full2x.size());
    float* d_pts = (float*)dev(pts.data(), pts.size() * 4);
    const size_t Pp = ((size_t)P + 127) / 128 * 128;
    float* d_out_sdf = (float*)dev(nullptr, Pp * 4);
    float* d_out_nrm = (float*)dev(nullptr, Pp * 12);
    void* d_feat = dev(nullptr, Pp * 256 * 2);
    void* d_S = dev(nullptr, Pp * 256 * 2 * 8);
    void* d_H = dev(nullptr, Pp * 256 * 2 * 8);
    void* d_V = dev(nullptr, Pp * 256 * 2 * 8);
    void* d_PE = dev(nullptr, Pp * 64 * 2);
    float* d_w8 = (float*)dev(lin[8].w.data(), 256 * 4);
    hipStream_t st;
    CK(hipStreamCreate(&st));

    VdnSdfArgs A;
    memset(&A, 0, sizeof(A));
    A.pts = d_pts; A.P = P; A.scale = 1.0f; A.sdf = d_out_sdf; A.feat = d_feat; A.normals = d_out_nrm; A.S = d_S; A.w8row = d_w8;
    A.n_per_ray = 1; A.z_ld = 1; A.sdf_ld = 1;

    struct Variant { std::string name; std::function<void()> run; double flop; };
    std::vector<Variant> vs;
    const double F1 = 1967104.0 * P, F0 = 918016.0 * P;
    auto with = [&](const char* blob, void* H, void* V, void* PE) { VdnSdfArgs a = A; a.blob = blob; a.H = H; a.V = V; a.PE = PE; return a; };
#define V2(TAG, M, S) vs.push_back({std::string("v2 mode") + #M + (S ? " training saves " : " ") + #TAG, [&] { \
        const bool x16 = std::string(#TAG).find("s16") != std::string::npos; \
        VdnSdfArgs a = M == 0 ? with(x16 ? d_sdf2x : d_sdf2, nullptr, nullptr, nullptr) : (S ? with(x16 ? d_full2x : d_full2, d_H, d_V, d_PE) : with(x16 ? d_full2x : d_full2, nullptr, nullptr, nullptr)); \
        if (std::string(#TAG).rfind("st_", 0) == 0 && !S) a.PE = d_PE; \
        sdf2_launch_##TAG(&a, st); }, M == 0 ? F0 : F1});
#include "sdf2_variants.inc"
#undef V2

    // ---- correctness: v2 mode 1 against the host evaluation on a sample of points ---------------------------------
    auto host_eval = [&](const float* x3, double& sdf, double* nrm, std::vector<double>& feat) {
        double pe[39];
        for (int d = 0; d < 3; ++d) pe[d] = x3[d];
        for (int k = 0; k < 6; ++k) for (int d = 0; d < 3; ++d) { pe[3 + 6 * k + d] = std::sin(x3[d] * (double)(1 << k)); pe[3 + 6 * k + 3 + d] = std::cos(x3[d] * (double)(1 << k)); }
        std::vector<std::vector<double>> xs(9), ss(8);
        xs[0].assign(pe, pe + 39);
        for (int l = 0; l < 8; ++l) {
            std::vector<double> in = xs[l];
            if (l == 4) { in.insert(in.end(), pe, pe + 39); for (auto& v : in) v /= std::sqrt(2.0); }
            xs[l + 1].resize(dout[l]); ss[l].resize(dout[l]);
            for (int r = 0; r < dout[l]; ++r) {
                double a = lin[l].b[r];
                for (int c = 0; c < din[l]; ++c) a += (double)lin[l].w[(size_t)r * din[l] + c] * in[c];
                const double z = 100.0 * a;
                xs[l + 1][r] = z > 20.0 ? a : std::log1p(std::exp(z)) / 100.0;
                ss[l][r] = 1.0 / (1.0 + std::exp(-z));
            }
        }
        feat.resize(256);
        for (int r = 0; r < 257; ++r) {
            double a = lin[8].b[r];
            for (int c = 0; c < 256; ++c) a += (double)lin[8].w[(size_t)r * 256 + c] * xs[8][c];
            if (r == 0) sdf = a; else feat[r - 1] = a;
        }
        std::vector<double> u(256), dpe(39, 0.0);
        for (int c = 0; c < 256; ++c) u[c] = lin[8].w[c];
        for (int l = 7; l >= 0; --l) {
            std::vector<double> v(dout[l]);
            for (int r = 0; r < dout[l]; ++r) v[r] = u[r] * ss[l][r];
            std::vector<double> un(din[l], 0.0);
            for (int r = 0; r < dout[l]; ++r) for (int c = 0; c < din[l]; ++c) un[c] += (double)lin[l].w[(size_t)r * din[l] + c] * v[r];
            if (l == 4) {
                for (auto& q : un) q /= std::sqrt(2.0);
                for (int i = 0; i < 39; ++i) dpe[i] += un[217 + i];
                un.resize(217);
            }
            if (l == 0) for (int i = 0; i < 39; ++i) dpe[i] += un[i];
            u = un;
        }
        for (int d = 0; d < 3; ++d) {
            nrm[d] = dpe[d];
            for (int k = 0; k < 6; ++k) {
                const double f = (double)(1 << k);
                nrm[d] += f * (std::cos(x3[d] * f) * dpe[3 + 6 * k + d] - std::sin(x3[d] * f) * dpe[3 + 6 * k + 3 + d]);
            }
        }
    };
    auto check = [&](const char* name, bool full) {
        CK(hipStreamSynchronize(st));
        std::vector<float> o_sdf(P), o_n((size_t)P * 3);
        std::vector<unsigned short> o_f(Pp * 256);
        CK(hipMemcpy(o_sdf.data(), d_out_sdf, (size_t)P * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o_n.data(), d_out_nrm, (size_t)P * 12, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o_f.data(), d_feat, Pp * 512, hipMemcpyDeviceToHost));
        double e_sdf = 0, e_n = 0, e_f = 0, m_n = 0, m_f = 0, cosmin = 1;
        const int NCHK = 96;
        for (int i = 0; i < NCHK; ++i) {
            const int p = (int)(((long)i * 7919 * 13 + (i % 3) * 31) % P);
            double sdf, nrm[3]; std::vector<double> feat;
            host_eval(&pts[(size_t)p * 3], sdf, nrm, feat);
            e_sdf = std::max(e_sdf, std::fabs(sdf - o_sdf[p]));
            if (full) {
                double dot = 0, na = 0, nb = 0;
                for (int d = 0; d < 3; ++d) { e_n = std::max(e_n, std::fabs(nrm[d] - o_n[(size_t)p * 3 + d])); m_n = std::max(m_n, std::fabs(nrm[d])); dot += nrm[d] * o_n[(size_t)p * 3 + d]; na += nrm[d] * nrm[d]; nb += (double)o_n[(size_t)p * 3 + d] * o_n[(size_t)p * 3 + d]; }
                cosmin = std::min(cosmin, dot / std::sqrt(na * nb + 1e-30));
                for (int f = 0; f < 256; ++f) {
                    const size_t idx = (size_t)(p >> 5) * (32 * 256) + (f >> 5) * 1024 + ((f >> 4) & 1) * 512 + ((f >> 2) & 1) * 256 + (p & 31) * 8 + ((f >> 3) & 1) * 4 + (f & 3);      // PT32 (mlp_engine.h)
                    e_f = std::max(e_f, std::fabs(feat[f] - bf2f(o_f[idx]))); m_f = std::max(m_f, std::fabs(feat[f]));
                }
            }
        }
        // FNV-1a over the raw output bits: variants that differ only in scheduling (ring depth, barrier cadence) must agree to the bit
        unsigned long long hsh = 1469598103934665603ull;
        auto eat = [&](const void* q, size_t n) { const unsigned char* c = (const unsigned char*)q; for (size_t i = 0; i < n; ++i) { hsh ^= c[i]; hsh *= 1099511628211ull; } };
        eat(o_sdf.data(), (size_t)P * 4);
        if (full) { eat(o_n.data(), (size_t)P * 12); eat(o_f.data(), (size_t)P * 512); }
        printf("check %-44s bits %016llx  sdf abs err %.3e", name, hsh, e_sdf);
        if (full) printf("  normal abs err %.3e (max |n| %.2f, min cosine %.6f)  feature abs err %.3e (max %.2f)", e_n, m_n, cosmin, e_f, m_f);
        printf("\n");
    };
    for (size_t i = 0; i < vs.size(); ++i) {
        CK(hipMemsetAsync(d_out_sdf, 0xff, (size_t)P * 4, st));
        CK(hipMemsetAsync(d_out_nrm, 0xff, (size_t)P * 12, st));
        vs[i].run();
        CK(hipGetLastError());
        check(vs[i].name.c_str(), vs[i].name.find("mode1") != std::string::npos);
    }

    // ---- timing: interleaved rounds, HIP events on the launch stream ----------------------------------------------
    std::vector<std::vector<float>> ms(vs.size());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) for (auto& v : vs) v.run();
    CK(hipStreamSynchronize(st));
    const int inner = 5;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < inner; ++k) vs[i].run();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            ms[i].push_back(t / inner);
        }
    // stamped variants (tag st_*): per-workgroup shader cycles and the clock they ran at
    for (size_t i = 0; i < vs.size(); ++i) {
        if (vs[i].name.find(" st_") == std::string::npos) continue;
        const int nwg = (P + 127) / 128;
        CK(hipMemsetAsync(d_PE, 0, (size_t)nwg * 64, st));
        for (int k = 0; k < 20; ++k) vs[i].run();          // warm, back to back: the clock the kernel holds under its own load
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> sb((size_t)nwg * 8);
        CK(hipMemcpy(sb.data(), d_PE, sb.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk, ph1, ph2;
        unsigned long long r_min = ~0ull, r_max = 0;
        for (int w = 0; w < nwg; ++w) {
            const double dc = (double)(sb[8 * w + 1] - sb[8 * w]), dr = (double)(sb[8 * w + 3] - sb[8 * w + 2]);
            cyc.push_back(dc); clk.push_back(dc / dr * 0.1);
            r_min = std::min(r_min, sb[8 * w + 2]); r_max = std::max(r_max, sb[8 * w + 3]);
            if (sb[8 * w + 4]) ph1.push_back((double)(sb[8 * w + 4] - sb[8 * w]));
            if (sb[8 * w + 5]) ph2.push_back((double)(sb[8 * w + 5] - sb[8 * w + 4]));
        }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        printf("%-44s per-WG cycles median %.0f (min %.0f max %.0f), clock median %.3f GHz, launch span %.1f us\n", vs[i].name.c_str(),
               cyc[nwg / 2], cyc[0], cyc[nwg - 1], clk[nwg / 2], (double)(r_max - r_min) * 0.01);
        if (!ph1.empty() && !ph2.empty()) {
            std::sort(ph1.begin(), ph1.end()); std::sort(ph2.begin(), ph2.end());
            printf("    phases: hidden layers (63 steps) %.0f cycles = %.0f / step, last layer (9 steps) %.0f = %.0f / step, sweep (59 steps) %.0f = %.0f / step\n",
                   ph1[ph1.size() / 2], ph1[ph1.size() / 2] / 63, ph2[ph2.size() / 2], ph2[ph2.size() / 2] / 9,
                   cyc[nwg / 2] - ph1[ph1.size() / 2] - ph2[ph2.size() / 2], (cyc[nwg / 2] - ph1[ph1.size() / 2] - ph2[ph2.size() / 2]) / 59);
        }
    }
    for (size_t i = 0; i < vs.size(); ++i) {
        std::sort(ms[i].begin(), ms[i].end());
        const double med = ms[i][ms[i].size() / 2], mn = ms[i][0];
        printf("%-44s median %8.1f us  min %8.1f us  %7.1f TFLOP/s  = %.3f of 2.5 PFLOP/s\n", vs[i].name.c_str(), med * 1e3, mn * 1e3, vs[i].flop / (med * 1e-3) / 1e12, vs[i].flop / (med * 1e-3) / 2.5e15);
    }
    return 0;
}
