// The comb table of a registered P-256 public key, built ON THE DEVICE (round 6).
//
// What it replaces: p256_tables29.h::build_comb_table<8> on a host thread - 6 ms per key, and a channel's first block makes all its
// endorsers eligible at once: eight of the fifteen milliseconds of a fresh provider's first pass were six of these (the reference pays the
// analogous cost when an identity first enters msp/cache/cache.go:14-18 through msp/mspimpl.go:402-426 - deserialization and KeyImport).
// The table itself is unchanged and BYTE-IDENTICAL to the host builder's (tests/test_gpu_parity.py compares them): CombTab<8>,
// T[w][d] = d * 2^(8 w) * Q for w = 0 .. 31, d = 1 .. 255, affine, fe29 Montgomery form of the canonical residue, 80-byte entries
// x[9] y[9] pad[2]; entry 0 of a window is zero.  640 KiB per key.
//
// Three launches for a batch of keys, none of them shaped like a GEMM (256-bit modular arithmetic on the integer VALU, as everywhere):
//   chain    one lane per key: B_w = 2^(8 w) Q, w = 0 .. 31 - 248 dependent doublings, the critical path (0.6 ms whatever the batch);
//   affine   one lane per (key, window): B_w to affine (safegcd inversion mod p, modinv30.h), so that the entries need mixed additions only;
//   entries  one lane per (key, window, digit): d * B_w by double-and-add from the affine base (<= 7 doublings, <= 7 mixed additions),
//            its own inversion, the canonical residue, and a coalesced 80-byte store (consecutive lanes = consecutive digits of a window).
// Exceptional cases cannot occur: Q has prime order n > 2^255 and every intermediate is k B_w with 2 <= k <= 255 plus or doubled.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "kernels.h"
#include "p256_verify29.h"

namespace fab {

struct KeyBaseAffine {
    fe x, y;   // normalised Montgomery form of the canonical residues
};

namespace {

// Jacobian (limb bounds of a point operation's output) -> affine, normalised: x = X / Z^2, y = Y / Z^3 as fe_to_mont of the canonical integers
__device__ void jac_to_affine_canon(fe& x, fe& y, const jac29& p) {
    const modinv_info PI = MODINV_P_INFO;
    u256 z, zi, xp, yp;
    fe_from_mont(z, p.Z);
    modinv(zi, z, PI);
    fe zim, zi2, zi3, xm, ym;
    fe_to_mont(zim, zi);
    fe_sqr(zi2, zim);
    fe_mul(zi3, zi2, zim);
    fe_mul(xm, p.X, zi2);
    fe_mul(ym, p.Y, zi3);
    fe_from_mont(xp, xm);          // the unique integers in [0, p) ...
    fe_from_mont(yp, ym);
    fe_to_mont(x, xp);             // ... in the representation the host builder stores (p256_tables29.h: fe_to_mont of the plain coordinate)
    fe_to_mont(y, yp);
}

}  // namespace

template <class Tab>
__global__ void __launch_bounds__(64) keytab_chain_kernel(uint32_t n_keys, const uint8_t* __restrict__ qxy, jac29* __restrict__ bases) {
    constexpr int BITS = 256 / Tab::WINDOWS;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_keys) return;
    u256 qx, qy;
    from_be32(qx, qxy + 64 * (size_t)k);
    from_be32(qy, qxy + 64 * (size_t)k + 32);
    jac29 acc;
    fe_to_mont(acc.X, qx);
    fe_to_mont(acc.Y, qy);
    fe_set_one(acc.Z);
    bases[(size_t)k * Tab::WINDOWS] = acc;
    for (int w = 1; w < Tab::WINDOWS; w++) {
        for (int b = 0; b < BITS; b++) {
            jac29 t;
            pt_dbl29(t, acc);
            acc = t;
        }
        bases[(size_t)k * Tab::WINDOWS + w] = acc;
    }
}

__global__ void __launch_bounds__(64) keytab_affine_kernel(uint32_t n_bases, const jac29* __restrict__ bases, KeyBaseAffine* __restrict__ aff) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bases) return;
    const jac29 b = bases[i];
    KeyBaseAffine a;
    jac_to_affine_canon(a.x, a.y, b);
    aff[i] = a;
}

template <class Tab>
__global__ void __launch_bounds__(256) keytab_entries_kernel(uint32_t n_keys, const KeyBaseAffine* __restrict__ aff, int32_t* const* __restrict__ tabs) {
    constexpr int BITS = 256 / Tab::WINDOWS;
    constexpr uint32_t PER_KEY = (uint32_t)Tab::WINDOWS << BITS;
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t key = id / PER_KEY, rem = id % PER_KEY, w = rem >> BITS, d = rem & ((1u << BITS) - 1u);
    if (key >= n_keys) return;
    int32_t* e = tabs[key] + Tab::index((int)w, d);
    if (d == 0) {
#pragma unroll
        for (int l = 0; l < COMB_ENTRY_WORDS; l++) e[l] = 0;
        return;
    }
    const KeyBaseAffine base = aff[(size_t)key * Tab::WINDOWS + w];
    fe x = base.x, y = base.y;
    if (d != 1) {
        jac29 acc;
        acc.X = base.x;
        acc.Y = base.y;
        fe_set_one(acc.Z);
        const int top = 31 - __clz((int)d);
        for (int b = top - 1; b >= 0; b--) {
            jac29 t;
            pt_dbl29(t, acc);
            acc = t;
            if ((d >> b) & 1u) {
                fe h, rr;
                pt_add_mixed29(t, acc, base.x, base.y, h, rr);
                acc = t;
            }
        }
        jac_to_affine_canon(x, y, acc);
    }
#pragma unroll
    for (int l = 0; l < 9; l++) {
        e[l] = x.v[l];
        e[9 + l] = y.v[l];
    }
    e[18] = 0;
    e[19] = 0;
}

template <class Tab>
static size_t comb_scratch_bytes(uint32_t n_keys) {
    return (size_t)n_keys * Tab::WINDOWS * (sizeof(jac29) + sizeof(KeyBaseAffine)) + 256;
}
size_t keytab_scratch_bytes(uint32_t n_keys) { return comb_scratch_bytes<KeyTab8>(n_keys); }
size_t gtab_scratch_bytes() { return comb_scratch_bytes<GTab16>(1) + 64 + sizeof(void*) + 64; }

template <class Tab>
static hipError_t launch_comb_build(uint32_t n_keys, const void* qxy, void* const* tabs, void* scratch, hipStream_t st) {
    if (n_keys == 0) return hipSuccess;
    constexpr int BITS = 256 / Tab::WINDOWS;
    jac29* bases = (jac29*)scratch;
    KeyBaseAffine* aff = (KeyBaseAffine*)(((uintptr_t)(bases + (size_t)n_keys * Tab::WINDOWS) + 15) & ~(uintptr_t)15);
    hipLaunchKernelGGL(keytab_chain_kernel<Tab>, dim3((n_keys + 63) / 64), dim3(64), 0, st, n_keys, (const uint8_t*)qxy, bases);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t nb = n_keys * (uint32_t)Tab::WINDOWS;
    hipLaunchKernelGGL(keytab_affine_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, nb, (const jac29*)bases, aff);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint64_t lanes = ((uint64_t)n_keys * Tab::WINDOWS) << BITS;
    hipLaunchKernelGGL(keytab_entries_kernel<Tab>, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, st, n_keys, (const KeyBaseAffine*)aff, (int32_t* const*)tabs);
    return hipGetLastError();
}

// qxy: n_keys x 64 bytes (X || Y, big-endian) on the device; tabs: n_keys device pointers (on the device) to tables of KeyTab8::TABLE_WORDS
// words each; scratch: keytab_scratch_bytes(n_keys).  Every key must be an affine point of the curve (the callers' gate).
hipError_t launch_keytab_build(uint32_t n_keys, const void* qxy, void* const* tabs, void* scratch, hipStream_t st) {
    return launch_comb_build<KeyTab8>(n_keys, qxy, tabs, scratch, st);
}

// A registered key's 16-bit comb (FABGPU_FLAG_KEY_TABLES_16BIT): the generator's format for any base point, same three launches.
size_t keytab16_scratch_bytes() { return comb_scratch_bytes<GTab16>(1); }
hipError_t launch_keytab16_build(uint32_t n_keys, const void* qxy, void* const* tabs, void* scratch, hipStream_t st) {
    return launch_comb_build<GTab16>(n_keys, qxy, tabs, scratch, st);
}

// The GENERATOR's comb (CombTab<16>: 16 windows x 65 535 affine points, 80 MiB) built the same way at fabgpu_init: one million lanes of
// at most fifteen doublings and fifteen mixed additions each - 2-3 ms of kernels against 0.2 s on sixteen host threads plus an 80 MiB
// upload (and every test context of the GPU suite used to pay that).  d_tab: GTab16::TABLE_WORDS words of device memory; scratch:
// gtab_scratch_bytes() of device memory (its tail holds G's coordinates and the table pointer, which this function uploads).
hipError_t launch_gtab_build(void* d_tab, void* scratch, hipStream_t st) {
    // (the generator's affine coordinates as plain integers, least significant word first: SEC 2 secp256r1; the same literals as p256_tables.h)
    const u256 gx = {{0xD898C296u, 0xF4A13945u, 0x2DEB33A0u, 0x77037D81u, 0x63A440F2u, 0xF8BCE6E5u, 0xE12C4247u, 0x6B17D1F2u}};
    const u256 gy = {{0x37BF51F5u, 0xCBB64068u, 0x6B315ECEu, 0x2BCE3357u, 0x7C0F9E16u, 0x8EE7EB4Au, 0xFE1A7F9Bu, 0x4FE342E2u}};
    struct Tail {
        uint8_t qxy[64];
        void* tab;
    } tail;
    to_be32(tail.qxy, gx);
    to_be32(tail.qxy + 32, gy);
    tail.tab = d_tab;
    uint8_t* d_tail = (uint8_t*)scratch + ((comb_scratch_bytes<GTab16>(1) + 63) & ~(size_t)63);
    hipError_t e = hipMemcpyAsync(d_tail, &tail, sizeof(tail), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(st);                              // (`tail` lives on this stack frame)
    if (e != hipSuccess) return e;
    return launch_comb_build<GTab16>(1, d_tail, (void* const*)(d_tail + offsetof(Tail, tab)), scratch, st);
}

// see warm_kernel_functions_kernels (kernels.hip)
int warm_kernel_functions_keytab() {
    int ok = 0;
    hipFuncAttributes a;
    const void* fns[] = {(const void*)keytab_chain_kernel<KeyTab8>, (const void*)keytab_affine_kernel, (const void*)keytab_entries_kernel<KeyTab8>,
                         (const void*)keytab_chain_kernel<GTab16>, (const void*)keytab_entries_kernel<GTab16>};
    for (const void* f : fns) ok += hipFuncGetAttributes(&a, f) == hipSuccess ? 1 : 0;
    return ok;
}

}  // namespace fab
