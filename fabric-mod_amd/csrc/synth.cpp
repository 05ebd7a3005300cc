// Synthetic block generator: random P-256 keypairs and low-S ECDSA signatures for benchmarks and parity tests
// (the role cryptogen / common/ledger/testutil/test_helper.go:270-279 play for the reference's tests).
// Host code over the same fp256/p256_point headers as the kernels; independent implementations (the C/Python
// oracle and OpenSSL) check what it emits in tests/test_host_logic.py.
#include <string.h>

#include <thread>
#include <vector>

#include "fabgpu_testhooks.h"
#include "p256_tables.h"

using namespace fab;

namespace {

struct Rng {  // xoshiro256**, seeded with splitmix64
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto& v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    void scalar(u256& k) {  // uniform in [1, 2^255): below n, never zero
        for (int i = 0; i < 4; i++) { uint64_t v = next(); k.w[2 * i] = (uint32_t)v; k.w[2 * i + 1] = (uint32_t)(v >> 32); }
        k.w[7] &= 0x7FFFFFFFu;
        if (is_zero(k)) k.w[0] = 1;
    }
};

const uint32_t* gtab() {
    static std::vector<uint32_t> tab = [] { std::vector<uint32_t> t(G_TABLE_WORDS); build_g_comb_table(t.data()); return t; }();
    return tab.data();
}

// k*G -> affine plain coordinates
void base_mul(const u256& k, u256& x, u256& y) {
    FlatGTab gt{gtab()};
    const u256 ONE = FAB_P256_R1;
    jac S; S.X = ONE; S.Y = ONE; S.Z = ONE;
    bool inf = true;
    for (int i = 0; i < G_WINDOWS; i++) {
        uint32_t d = nibble(k, i);
        if (!d) continue;
        u256 gx, gy;
        gt.load(i, d, gx, gy);
        if (inf) { S.X = gx; S.Y = gy; S.Z = ONE; inf = false; continue; }
        jac t; bool hz, rz;
        pt_add_mixed(t, S, gx, gy, hz, rz);
        S = t;
    }
    u256 mx, my;
    jac_to_affine_mont(mx, my, S);
    fp_from_mont(x, mx);
    fp_from_mont(y, my);
}

void reduce_n(u256& r, const u256& a) {
    const u256 N = FAB_P256_N;
    u256 t;
    uint32_t br = sub256(t, a, N);
    sel256(r, br == 0, t, a);
}

void make_one(Rng& rng, const uint8_t* e_in, uint8_t* qx, uint8_t* qy, uint8_t* e_out, uint8_t* r_out, uint8_t* s_out) {
    const u256 N = FAB_P256_N, HALF = FAB_P256_HALF_N;
    for (;;) {
        u256 d, k, e, x, y, r, s;
        rng.scalar(d);
        rng.scalar(k);
        if (e_in) from_be32(e, e_in);
        else rng.scalar(e), e.w[7] |= (uint32_t)(rng.next() & 0x80000000u);
        base_mul(d, x, y);
        to_be32(qx, x);
        to_be32(qy, y);
        base_mul(k, x, y);
        reduce_n(r, x);
        if (is_zero(r)) continue;
        // s = k^-1 (e + r d) mod n   (bccsp/sw/ecdsa.go:27-39 signECDSA incl. utils.ToLowS)
        u256 km, ki, rm, dm, em, t, ered;
        fn_to_mont(km, k); fn_inv(ki, km);
        fn_to_mont(rm, r); fn_to_mont(dm, d);
        reduce_n(ered, e); fn_to_mont(em, ered);
        fn_mul(t, rm, dm); fn_add(t, t, em); fn_mul(t, t, ki); fn_from_mont(s, t);
        if (is_zero(s)) continue;
        if (lt256(HALF, s)) sub256(s, N, s);
        to_be32(e_out, e);
        to_be32(r_out, r);
        to_be32(s_out, s);
        return;
    }
}

}  // namespace

extern "C" int fabgpu_synth_batch(size_t n, uint64_t seed, uint32_t invalid_permille, const uint8_t* e_in, uint8_t* qx, uint8_t* qy,
                                  uint8_t* e_out, uint8_t* r, uint8_t* s, uint8_t* kind, int threads) {
    if (n && (!qx || !qy || !e_out || !r || !s)) return FABGPU_EINVAL;
    if (invalid_permille > 1000) return FABGPU_EINVAL;
    gtab();
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    const size_t CH = 256;  // chunk i gets its own stream: output independent of the thread count
    size_t nch = (n + CH - 1) / CH;
    auto work = [&](int tid) {
        for (size_t c = tid; c < nch; c += threads) {
            Rng rng(seed * 0x100000001B3ull + c);
            size_t lo = c * CH, hi = lo + CH < n ? lo + CH : n;
            for (size_t i = lo; i < hi; i++)
                make_one(rng, e_in ? e_in + 32 * i : nullptr, qx + 32 * i, qy + 32 * i, e_out + 32 * i, r + 32 * i, s + 32 * i);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    // mutations (sequential, deterministic)
    if (kind) memset(kind, 0, n);
    size_t nbad = (size_t)((uint64_t)n * invalid_permille / 1000);
    if (nbad) {
        Rng rng(seed ^ 0xBADC0FFEE0DDF00Dull);
        std::vector<uint8_t> used(n, 0);
        const u256 N = FAB_P256_N;
        for (size_t j = 0; j < nbad; j++) {
            size_t i;
            do { i = (size_t)(rng.next() % n); } while (used[i]);
            used[i] = 1;
            uint8_t m = (uint8_t)(1 + j % 4);
            if (kind) kind[i] = m;
            if (m == 1) {
                e_out[32 * i + (rng.next() % 32)] ^= (uint8_t)(1u << (rng.next() % 8));
            } else if (m == 2) {
                size_t o = (i + 1) % n;
                if (o == i) { qy[32 * i + 31] ^= 1; }
                else { memcpy(qx + 32 * i, qx + 32 * o, 32); memcpy(qy + 32 * i, qy + 32 * o, 32); }
            } else if (m == 3) {
                u256 sv; from_be32(sv, s + 32 * i); sub256(sv, N, sv); to_be32(s + 32 * i, sv);
            } else {
                u256 rv, one = zero256(); one.w[0] = 1;
                from_be32(rv, r + 32 * i); add256(rv, rv, one); to_be32(r + 32 * i, rv);
            }
        }
    }
    return FABGPU_OK;
}
