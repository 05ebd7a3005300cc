// HOST-ONLY TEST HOOKS - built into libfabgpu_hosttest.so, never into the product library libfabgpu.so.
// They run the SAME header code the kernels are compiled from (fp256.h / p256_point.h / p256_tables.h) on
// the CPU so that the arithmetic and the verification core can be unit-tested in a container without a GPU.
// Nothing in the product path links or loads this file.
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#define FE29_CHECK 1
#include "bn_tables29.h"
#include "p256_tables29.h"
#include "multi_plan.h"
#include "coalescer.h"
#include "pass_route.h"
#include "block_walk_core.h"

using namespace fab;

static const uint32_t* gtab() {
    static std::vector<uint32_t> tab = [] { std::vector<uint32_t> t(G_TABLE_WORDS); build_g_comb_table(t.data()); return t; }();
    return tab.data();
}

static const int32_t* gtab29() {
    static std::vector<int32_t> tab = [] { std::vector<int32_t> t(GTab16::TABLE_WORDS); build_g_comb_table16(t.data()); return t; }();
    return tab.data();
}

extern "C" {

// fe29 field ops on plain integers (32 big-endian bytes in, out): the op is carried out in the Montgomery domain
// op: 0 mul 1 sqr 2 add 3 sub 4 (a+b)*(a-b) with lazy operands 5 is_zero(a-b) 6 round trip
void hosttest_fe29_op(int op, const uint8_t* a32, const uint8_t* b32, uint8_t* out32) {
    u256 a, b, r = zero256();
    from_be32(a, a32);
    from_be32(b, b32);
    fe fa, fb, fr, t1, t2;
    fe_to_mont(fa, a);
    fe_to_mont(fb, b);
    switch (op) {
        case 0: fe_mul(fr, fa, fb); break;
        case 1: fe_sqr(fr, fa); break;
        case 2: fe_add(fr, fa, fb); break;
        case 3: fe_sub(fr, fa, fb); break;
        case 4: fe_add(t1, fa, fb); fe_sub(t2, fa, fb); fe_mul(fr, t1, t2); break;
        case 5: fe_sub(t1, fa, fb); fr = fa; r.w[0] = fe_is_zero(t1) ? 1 : 0; to_be32(out32, r); return;
        case 6: fr = fa; break;   // to_mont / from_mont round trip (any 256-bit input)
        default: fr = fa;
    }
    fe_from_mont(r, fr);
    to_be32(out32, r);
}
// which: 0 = mod n, 1 = mod p
void hosttest_modinv(int which, const uint8_t* a32, uint8_t* out32) {
    const modinv_info NI = MODINV_N_INFO, PI = MODINV_P_INFO;
    u256 a, r;
    from_be32(a, a32);
    modinv(r, a, which ? PI : NI);
    to_be32(out32, r);
}
// modinv_divsteps30_column (what a lane of the pair kernels runs) against modinv_divsteps30: returns 0 when the two columns are the
// full matrix's and the zetas agree
int hosttest_divsteps_columns(int32_t zeta, uint32_t f0, uint32_t g0) {
    trans2x2 t;
    const int32_t z = modinv_divsteps30(zeta, f0, g0, t);
    int32_t u = 1, q = 0, v = 0, r = 1;
    const int32_t z1 = modinv_divsteps30_column(zeta, f0, g0, u, q);
    const int32_t z2 = modinv_divsteps30_column(zeta, f0, g0, v, r);
    return (z == z1 && z == z2 && u == t.u && q == t.q && v == t.v && r == t.r) ? 0 : 1;
}
void hosttest_gtab29_entry(int window, int digit, uint8_t* x32, uint8_t* y32) {
    GTab16 gt{gtab29()};
    fe x, y;
    u256 px, py;
    gt.load(window, (uint32_t)digit, x, y);
    fe_from_mont(px, x);
    fe_from_mont(py, y);
    to_be32(x32, px);
    to_be32(y32, py);
}
// R = u1*G + u2*Q through the kernel's CombinedMult (arbitrary scalars < n, u2 != 0): affine x, y out; returns 1 for infinity
int hosttest_combined_mult29(const uint8_t* u1_32, const uint8_t* u2_32, const uint8_t* qx32, const uint8_t* qy32, uint8_t* x32, uint8_t* y32) {
    GTab16 gt{gtab29()};
    const fe ONE = {FE29_R1};
    u256 u1, u2, qx, qy;
    from_be32(u1, u1_32); from_be32(u2, u2_32); from_be32(qx, qx32); from_be32(qy, qy32);
    jac29 Q, R;
    fe_to_mont(Q.X, qx);
    fe_to_mont(Q.Y, qy);
    Q.Z = ONE;
    LocalQTab29 qtab;
    bool inf;
    p256_combined_mult29(R, inf, u1, u2, Q, gt, qtab);
    if (inf) return 1;
    // affine: x = X / Z^2, y = Y / Z^3 with the safegcd inversion mod p
    const modinv_info PI = MODINV_P_INFO;
    u256 zp, zi, x, y;
    fe_from_mont(zp, R.Z);
    modinv(zi, zp, PI);
    fe fzi, zi2, zi3, fx, fy;
    fe_to_mont(fzi, zi);
    fe_sqr(zi2, fzi);
    fe_mul(zi3, zi2, fzi);
    fe_weak_norm(R.Y, R.Y);
    fe_mul(fx, R.X, zi2);
    fe_mul(fy, R.Y, zi3);
    fe_from_mont(x, fx);
    fe_from_mont(y, fy);
    to_be32(x32, x);
    to_be32(y32, y);
    return 0;
}
// the keyed core (registered public key: comb table of Q built like the device would) on the CPU
void hosttest_verify_keyed_core29(size_t n, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* e, const uint8_t* r, const uint8_t* s,
                                  uint8_t* status) {
    GTab16 gt{gtab29()};
    u256 qx, qy;
    from_be32(qx, qx32);
    from_be32(qy, qy32);
    std::vector<int32_t> kt(KeyTab8::TABLE_WORDS);
    build_key_comb_table8(kt.data(), qx, qy);
    KeyTab8 kk{kt.data()};
    for (size_t i = 0; i < n; i++) {
        u256 ve, vr, vs;
        from_be32(ve, e + 32 * i); from_be32(vr, r + 32 * i); from_be32(vs, s + 32 * i);
        status[i] = (uint8_t)p256_verify_keyed_core29(ve, vr, vs, gt, kk);
    }
}
void hosttest_verify_core29(size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* e, const uint8_t* r, const uint8_t* s,
                            uint8_t* status) {
    GTab16 gt{gtab29()};
    for (size_t i = 0; i < n; i++) {
        u256 vqx, vqy, ve, vr, vs;
        from_be32(vqx, qx + 32 * i); from_be32(vqy, qy + 32 * i); from_be32(ve, e + 32 * i);
        from_be32(vr, r + 32 * i); from_be32(vs, s + 32 * i);
        LocalQTab29 qtab;
        status[i] = (uint8_t)p256_verify_core29(vqx, vqy, ve, vr, vs, gt, qtab);
    }
}

// FAKE MULTI-DEVICE BACKEND ("G host threads running the CPU backend + a memcpy all-gather"): the partition, the
// equal-count exchange and the final layout of fabgpu_multi.hip on a box without GPUs.  Every "device" is a host thread that runs the
// kernel's verification core (the same header code) over its shard and packs the verdicts into shard words like the wave ballot does;
// the "all-gather" concatenates the G x words_per_rank words on every rank; rank 0's copy is laid out as one bitmap.
// off == NULL: shards by count; else by message bytes (the digests e are still given: hashing is not this test's subject).
// merged_check: set to 1 if every rank ended up with the same gathered buffer.
int hosttest_multi_verify(size_t n, uint32_t G, const uint32_t* off, const uint8_t* qx, const uint8_t* qy, const uint8_t* e, const uint8_t* r,
                          const uint8_t* s, uint64_t* verdict_bits, uint8_t* status, int* merged_check) {
    if (!G) return -1;
    const ShardPlan p = off ? plan_by_bytes(n, off, G) : plan_by_count(n, G);
    const size_t wpr = p.words_per_rank;
    std::vector<std::vector<uint64_t>> words(G, std::vector<uint64_t>(wpr, 0));
    gtab29();                                                   // built once, before the threads race for it
    std::vector<std::thread> th;
    for (uint32_t g = 0; g < G; g++)
        th.emplace_back([&, g] {
            const size_t lo = p.lo[g], cnt = p.hi[g] - p.lo[g];
            if (!cnt) return;
            std::vector<uint8_t> st(cnt);
            hosttest_verify_core29(cnt, qx + 32 * lo, qy + 32 * lo, e + 32 * lo, r + 32 * lo, s + 32 * lo, st.data());
            for (size_t i = 0; i < cnt; i++) {
                if (st[i] == 0) words[g][i >> 6] |= (uint64_t)1 << (i & 63);
                if (status) status[lo + i] = st[i];
            }
        });
    for (auto& t : th) t.join();
    // the all-gather: every rank receives rank 0's, rank 1's, ... words in rank order
    std::vector<std::vector<uint64_t>> merged(G, std::vector<uint64_t>((size_t)G * wpr));
    for (uint32_t dst = 0; dst < G; dst++)
        for (uint32_t src = 0; src < G; src++)
            if (wpr) memcpy(merged[dst].data() + (size_t)src * wpr, words[src].data(), wpr * 8);
    int same = 1;
    for (uint32_t dst = 1; dst < G; dst++) same = same && merged[dst] == merged[0];
    if (merged_check) *merged_check = same;
    const size_t total_words = (n + 63) / 64;
    for (uint32_t g = 0; g < G; g++) {
        const size_t cnt = p.hi[g] - p.lo[g];
        if (!cnt) continue;
        const size_t w = (cnt + 63) / 64;
        if (p.word_at[g] + w > total_words) return -2;
        memcpy(verdict_bits + p.word_at[g], merged[0].data() + (size_t)g * wpr, w * 8);
    }
    return 0;
}

// The coalescer of one-signature calls (coalescer.h) with a fake device: `threads` callers make `calls` calls each; a "launch" takes
// launch_us and answers request x with 3 * x + 1.  Returns the number of wrong answers (0 expected); every caller must get its own.
// Also reports how many launches served the calls and the largest batch - with callers arriving while a launch runs, far fewer
// launches than calls.  fail_every > 0: every fail_every-th launch "fails" (answers -1 for the whole batch, as a device error would).
int hosttest_coalescer(int threads, int calls, uint32_t launch_us, uint32_t window_us, uint32_t max_batch, int fail_every, uint64_t* launches,
                       uint64_t* largest, uint64_t* failed_answers) {
    struct Req : CoalescedBase {
        int64_t x, y;
    };
    Coalescer<Req> co;
    co.configure(window_us, max_batch);
    std::atomic<int> wrong(0), failed(0), launch_no(0);
    std::atomic<int> over(0);
    auto runner = [&](std::vector<Req*>& batch) {
        if (batch.size() > max_batch) over++;
        const int k = ++launch_no;
        if (launch_us) std::this_thread::sleep_for(std::chrono::microseconds(launch_us));
        const bool fail = fail_every > 0 && k % fail_every == 0;
        for (Req* q : batch) q->y = fail ? -1 : 3 * q->x + 1;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            for (int c = 0; c < calls; c++) {
                Req r;
                r.x = (int64_t)t * 1000003 + c;
                r.y = 0;
                co.submit(&r, runner);
                if (r.y == -1) failed++;
                else if (r.y != 3 * r.x + 1) wrong++;
            }
        });
    for (auto& t : th) t.join();
    uint64_t c = 0;
    co.stats(&c, launches, largest);
    if (failed_answers) *failed_answers = (uint64_t)failed.load();
    if (c != (uint64_t)threads * calls) return -1;
    if (over.load()) return -2;
    return wrong.load();
}

// op: 0 fp_mul 1 fp_sqr 2 fp_add 3 fp_sub 4 fp_to_mont 5 fp_from_mont 6 fn_mul 7 fn_sqr 8 fn_to_mont 9 fn_from_mont 10 fn_inv 11 fp_inv
void hosttest_fieldop(int op, const uint8_t* a32, const uint8_t* b32, uint8_t* out32) {
    u256 a, b, r;
    from_be32(a, a32);
    from_be32(b, b32);
    switch (op) {
        case 0: fp_mul(r, a, b); break;
        case 1: fp_sqr(r, a); break;
        case 2: fp_add(r, a, b); break;
        case 3: fp_sub(r, a, b); break;
        case 4: fp_to_mont(r, a); break;
        case 5: fp_from_mont(r, a); break;
        case 6: fn_mul(r, a, b); break;
        case 7: fn_sqr(r, a); break;
        case 8: fn_to_mont(r, a); break;
        case 9: fn_from_mont(r, a); break;
        case 10: fn_inv(r, a); break;
        case 11: fp_inv(r, a); break;
        default: r = zero256();
    }
    to_be32(out32, r);
}

// the verification core on the CPU, same template the kernel instantiates
void hosttest_verify_core(size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* e, const uint8_t* r, const uint8_t* s,
                          uint8_t* status) {
    FlatGTab gt{gtab()};
    for (size_t i = 0; i < n; i++) {
        u256 vqx, vqy, ve, vr, vs;
        from_be32(vqx, qx + 32 * i); from_be32(vqy, qy + 32 * i); from_be32(ve, e + 32 * i);
        from_be32(vr, r + 32 * i); from_be32(vs, s + 32 * i);
        jac qtab[16];
        status[i] = (uint8_t)p256_verify_core(vqx, vqy, ve, vr, vs, gt, qtab);
    }
}

// comb table entry (window, digit) in plain affine coordinates
void hosttest_gtab_entry(int window, int digit, uint8_t* x32, uint8_t* y32) {
    FlatGTab gt{gtab()};
    u256 x, y, px, py;
    gt.load(window, (uint32_t)digit, x, y);
    fp_from_mont(px, x);
    fp_from_mont(py, y);
    to_be32(x32, px);
    to_be32(y32, py);
}
// ---- FP256BN (idemix) ----
// op: 0 mul 1 sqr 2 (a+b)*(a-b) lazy 3 is_zero(a-b) 4 round trip 5 (3a)^2 6 (4a)*b
void hosttest_bn29_op(int op, const uint8_t* a32, const uint8_t* b32, uint8_t* out32) {
    u256 a, b, r = zero256();
    from_be32(a, a32);
    from_be32(b, b32);
    fbn fa, fb, fr, t1, t2;
    fe_to_mont(fa, a);
    fe_to_mont(fb, b);
    switch (op) {
        case 0: fe_mul(fr, fa, fb); break;
        case 1: fe_sqr(fr, fa); break;
        case 2: fe_add(t1, fa, fb); fe_sub(t2, fa, fb); fe_mul(fr, t1, t2); break;
        case 3: fe_sub(t1, fa, fb); r.w[0] = fe_is_zero(t1) ? 1 : 0; to_be32(out32, r); return;
        case 5: fe_add(t1, fa, fa); fe_add(t1, t1, fa); fe_sqr(fr, t1); break;
        case 6: fe_add(t1, fa, fa); fe_add(t1, t1, t1); fe_mul(fr, t1, fb); break;
        default: fr = fa;
    }
    fe_from_mont(r, fr);
    to_be32(out32, r);
}
void hosttest_bn_modinv(const uint8_t* a32, uint8_t* out32) {
    const modinv_info PI = MODINV_BNP_INFO;
    u256 a, r;
    from_be32(a, a32);
    modinv(r, a, PI);
    to_be32(out32, r);
}
struct BnIssuerTabs {
    std::vector<int32_t> hsk, hrand;
};
void* hosttest_bn_issuer_new(const uint8_t* hskx, const uint8_t* hsky, const uint8_t* hrx, const uint8_t* hry) {
    BnIssuerTabs* t = new BnIssuerTabs;
    t->hsk.resize(KeyTab8::TABLE_WORDS);
    t->hrand.resize(KeyTab8::TABLE_WORDS);
    u256 x, y;
    from_be32(x, hskx); from_be32(y, hsky);
    build_bn_comb_table8(t->hsk.data(), x, y);
    from_be32(x, hrx); from_be32(y, hry);
    build_bn_comb_table8(t->hrand.data(), x, y);
    return t;
}
void hosttest_bn_issuer_free(void* p) { delete (BnIssuerTabs*)p; }
// which: 0 HSk 1 HRand
void hosttest_bn_tab_entry(void* p, int which, int window, int digit, uint8_t* x32, uint8_t* y32) {
    BnIssuerTabs* t = (BnIssuerTabs*)p;
    KeyTab8 kt{which ? t->hrand.data() : t->hsk.data()};
    fbn x, y;
    u256 px, py;
    kt.load(window, (uint32_t)digit, x, y);
    fe_from_mont(px, x);
    fe_from_mont(py, y);
    to_be32(x32, px);
    to_be32(y32, py);
}
// the commitment t = HSk s_sk + HRand s_rnym - Nym c through the kernel's core; returns its status
int hosttest_bn_nym_commitment(void* p, const uint8_t* nx32, const uint8_t* ny32, const uint8_t* c32, const uint8_t* ssk32,
                               const uint8_t* srn32, uint8_t* tx32, uint8_t* ty32) {
    BnIssuerTabs* t = (BnIssuerTabs*)p;
    KeyTab8 hsk{t->hsk.data()}, hrand{t->hrand.data()};
    u256 nx, ny, c, ssk, srn, tx, ty;
    from_be32(nx, nx32); from_be32(ny, ny32); from_be32(c, c32); from_be32(ssk, ssk32); from_be32(srn, srn32);
    LocalQTab<fbn> qtab;
    uint32_t st = bn_nym_commitment29(tx, ty, nx, ny, c, ssk, srn, hsk, hrand, qtab);
    to_be32(tx32, tx);
    to_be32(ty32, ty);
    return (int)st;
}
// GLV decomposition of the device: k = +-m1 +- m2 lambda (mod r); flags bit0 = k1 negative, bit1 = k2 negative
int hosttest_bn_glv_decompose(const uint8_t* k32, uint8_t* m1_32, uint8_t* m2_32) {
    u256 k, m1, m2;
    bool n1, n2;
    from_be32(k, k32);
    bn_glv_decompose(m1, n1, m2, n2, k);
    to_be32(m1_32, m1);
    to_be32(m2_32, m2);
    return (n1 ? 1 : 0) | (n2 ? 2 : 0);
}
// +-m1 Q +- m2 phi(Q) through glv_mult29; with identity_beta the "endomorphism" is the identity, which makes collisions
// (accumulator == addend) constructible: the return value has bit0 = infinity, bit1 = collision reported.
int hosttest_bn_glv_mult(const uint8_t* qx32, const uint8_t* qy32, const uint8_t* m1_32, int n1, const uint8_t* m2_32, int n2, int identity_beta,
                         uint8_t* x32, uint8_t* y32) {
    u256 qx, qy, m1, m2;
    from_be32(qx, qx32); from_be32(qy, qy32); from_be32(m1, m1_32); from_be32(m2, m2_32);
    jacbn Q, T;
    fe_to_mont(Q.X, qx);
    fe_to_mont(Q.Y, qy);
    fe_set_one(Q.Z);
    fbn beta = {BN29_BETA_MONT};
    if (identity_beta) fe_set_one(beta);
    LocalQTab<fbn> qtab;
    bool inf, exc;
    glv_mult29(T, inf, exc, m1, n1 != 0, m2, n2 != 0, Q, beta, qtab);
    const modinv_info PI = MODINV_BNP_INFO;
    u256 zp, zi, x, y;
    fe_from_mont(zp, T.Z);
    modinv(zi, zp, PI);
    fbn zm, z2, z3, ax, ay;
    fe_to_mont(zm, zi);
    fe_sqr(z2, zm);
    fe_mul(z3, z2, zm);
    fe_mul(ax, T.X, z2);
    fe_mul(ay, T.Y, z3);
    fe_from_mont(x, ax);
    fe_from_mont(y, ay);
    to_be32(x32, x);
    to_be32(y32, y);
    return (inf ? 1 : 0) | (exc ? 2 : 0);
}
// the two-lanes-per-signature form of the commitment: both halves on the CPU, the exchange as plain copies
int hosttest_bn_nym_commitment_split(void* p, const uint8_t* nx32, const uint8_t* ny32, const uint8_t* c32, const uint8_t* ssk32,
                                     const uint8_t* srn32, uint8_t* tx32, uint8_t* ty32) {
    BnIssuerTabs* t = (BnIssuerTabs*)p;
    KeyTab8 hsk{t->hsk.data()}, hrand{t->hrand.data()};
    u256 nx, ny, c, ssk, srn, txe, tye, txo, tyo;
    from_be32(nx, nx32); from_be32(ny, ny32); from_be32(c, c32); from_be32(ssk, ssk32); from_be32(srn, srn32);
    LocalQTab<fbn> qe, qo;
    bn_nym_half he, ho;
    bn_nym_split_part1(he, false, nx, ny, c, ssk, srn, hsk, hrand, qe);
    bn_nym_split_part1(ho, true, nx, ny, c, ssk, srn, hsk, hrand, qo);
    uint32_t se = bn_nym_split_part2(txe, tye, he, ho.P, ho.inf);
    uint32_t so = bn_nym_split_part2(txo, tyo, ho, he.P, he.inf);
    to_be32(tx32, txe);
    to_be32(ty32, tye);
    if (se != so) return -1;                                               // the two lanes must agree on the status ...
    if (se == NYM_VALID && (!eq256(txe, txo) || !eq256(tye, tyo))) return -2;   // ... and, when it is meaningful, on t
    return (int)se;
}
// a 16-bit comb table of one base (what the device uses), built on the CPU: entry (window, digit) in plain affine coordinates.
// The table (80 MiB) is built once per (x, y) and cached for the following calls.
void hosttest_bn_tab16_entry(const uint8_t* bx32, const uint8_t* by32, int window, int digit, uint8_t* x32, uint8_t* y32) {
    static std::vector<int32_t> tab;
    static u256 cx = zero256(), cy = zero256();
    u256 bx, by;
    from_be32(bx, bx32);
    from_be32(by, by32);
    if (tab.empty() || !eq256(bx, cx) || !eq256(by, cy)) {
        tab.assign(GTab16::TABLE_WORDS, 0);
        build_bn_comb_table<16>(tab.data(), bx, by, 8);
        cx = bx;
        cy = by;
    }
    GTab16 kt{tab.data()};
    fbn x, y;
    u256 px, py;
    kt.load(window, (uint32_t)digit, x, y);
    fe_from_mont(px, x);
    fe_from_mont(py, y);
    to_be32(x32, px);
    to_be32(y32, py);
}
// the provider's pass routing (pass_route.h): which of n_devices contexts a block pass named block_seq goes to
int hosttest_route_block(uint64_t block_seq, const uint32_t* in_flight, int n_devices) { return route_block(block_seq, in_flight, n_devices); }
// The certificate walk of the device route over a WINDOW of the DER (block_walk_core.h cert_der_p256_key_offset_window): `avail` of the
// certificate's `len` bytes are readable.  Offset of X, -1 not a P-256 certificate, -2 the answer needs bytes beyond the window.
int hosttest_cert_key_offset_window(const uint8_t* der, size_t avail, size_t len) {
    return (int)bccsp::walk::cert_der_p256_key_offset_window(der, avail, len);
}
// the digest memo's slot choice (block_walk_core.h msg_fingerprint) over the message a || b
uint64_t fabgpu_hosttest_msg_fingerprint(const uint8_t* a, uint32_t alen, const uint8_t* b, uint32_t blen) {
    return bccsp::walk::msg_fingerprint(a, alen, b, blen);
}
}
