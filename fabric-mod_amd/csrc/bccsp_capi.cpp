// Flat C wrappers (include/fabgpu_bccsp.h) over the C++ host mirror, for ctypes / other FFIs.
#include <stdio.h>
#include <string.h>

#include "../../include/fabgpu_bccsp.h"
#include "bccsp_host.h"

using namespace fab::bccsp;

struct fabgpu_csp {
    std::unique_ptr<GPUCSP> csp;
};

namespace {
void put_err(char* dst, size_t cap, const std::string& s) {
    if (!dst || cap == 0) return;
    size_t k = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(dst, s.data(), k);
    dst[k] = 0;
}
}  // namespace

extern "C" {

int fabgpu_csp_new(const fabgpu_cfg* cfg, fabgpu_csp** out, char* err, size_t errcap) {
    if (!out) return FABGPU_EINVAL;
    *out = nullptr;
    fabgpu_csp* h = new fabgpu_csp();
    Error e = GPUCSP::New(cfg, h->csp);
    if (!e.ok()) {
        put_err(err, errcap, e.msg);
        delete h;
        return FABGPU_ENODEV;
    }
    put_err(err, errcap, "");
    *out = h;
    return FABGPU_OK;
}
void fabgpu_csp_free(fabgpu_csp* csp) { delete csp; }
fabgpu_ctx* fabgpu_csp_ctx(fabgpu_csp* csp) { return csp ? csp->csp->ctx() : nullptr; }

int fabgpu_csp_key_import(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, int* on_curve, char* err, size_t errcap) {
    if (!csp) return FABGPU_EINVAL;
    ECDSAPublicKey k;
    Error e = csp->csp->KeyImport(qx32, qy32, k, true);
    put_err(err, errcap, e.ok() ? "" : e.msg);
    if (on_curve) *on_curve = k.on_curve ? 1 : 0;
    return FABGPU_OK;
}

int fabgpu_csp_hash(fabgpu_csp* csp, const uint8_t* msg, size_t len, const char* alg, uint8_t* digest32, char* err, size_t errcap) {
    if (!csp || !digest32) return FABGPU_EINVAL;
    HashOpts o;
    if (alg) o.algorithm = alg;
    std::vector<uint8_t> d;
    Error e = csp->csp->Hash(msg, len, alg ? &o : nullptr, d);
    put_err(err, errcap, e.ok() ? "" : e.msg);
    if (e.ok()) memcpy(digest32, d.data(), 32);
    return FABGPU_OK;
}

int fabgpu_csp_verify(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* sig, size_t siglen,
                      const uint8_t* digest, size_t dlen, int* valid, int* flags, char* err, size_t errcap) {
    if (!csp || !valid) return FABGPU_EINVAL;
    ECDSAPublicKey k;
    const ECDSAPublicKey* kp = nullptr;
    if (qx32 && qy32) {
        csp->csp->KeyImport(qx32, qy32, k);
        kp = &k;
    }
    VerifyResult r = csp->csp->Verify(kp, sig, siglen, digest, dlen);
    if (r.infrastructure) {
        put_err(err, errcap, r.err.msg);
        return FABGPU_ELAUNCH;
    }
    *valid = r.valid ? 1 : 0;
    if (flags) *flags = r.needs_sw ? 1 : 0;
    put_err(err, errcap, r.err.ok() ? "" : r.err.msg);
    return FABGPU_OK;
}

int fabgpu_csp_verify_batch(fabgpu_csp* csp, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* sig_arena,
                            const uint32_t* sig_off, const uint8_t* dig_arena, const uint32_t* dig_off, uint8_t* valid,
                            char* errs, size_t errstride) {
    if (!csp || (n && (!qx || !qy || !sig_off || !dig_off || !valid))) return FABGPU_EINVAL;
    std::vector<ECDSAPublicKey> keys(n);
    std::vector<VerifyItem> items(n);
    for (size_t i = 0; i < n; i++) {
        csp->csp->KeyImport(qx + 32 * i, qy + 32 * i, keys[i]);
        items[i] = {&keys[i], sig_arena + sig_off[i], sig_off[i + 1] - sig_off[i], dig_arena + dig_off[i], dig_off[i + 1] - dig_off[i]};
    }
    std::vector<VerifyResult> res;
    Error e = csp->csp->VerifyBatch(items, res);
    if (!e.ok()) return FABGPU_ELAUNCH;
    for (size_t i = 0; i < n; i++) {
        valid[i] = res[i].valid ? 1 : 0;
        if (errs && errstride) put_err(errs + i * errstride, errstride, res[i].err.ok() ? "" : res[i].err.msg);
    }
    return FABGPU_OK;
}

int fabgpu_csp_identity_verify_batch(fabgpu_csp* csp, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* msg_arena,
                                     const uint32_t* msg_off, const uint8_t* sig_arena, const uint32_t* sig_off, char* errs,
                                     size_t errstride) {
    if (!csp || (n && (!qx || !qy || !msg_off || !sig_off || !errs || !errstride))) return FABGPU_EINVAL;
    std::vector<ECDSAPublicKey> keys(n);
    std::vector<IdentityItem> items(n);
    for (size_t i = 0; i < n; i++) {
        csp->csp->KeyImport(qx + 32 * i, qy + 32 * i, keys[i]);
        items[i] = {&keys[i], msg_arena + msg_off[i], msg_off[i + 1] - msg_off[i], sig_arena + sig_off[i], sig_off[i + 1] - sig_off[i]};
    }
    std::vector<std::string> out;
    Error e = csp->csp->IdentityVerifyBatch(items, out);
    if (!e.ok()) return FABGPU_ELAUNCH;
    for (size_t i = 0; i < n; i++) put_err(errs + i * errstride, errstride, out[i]);
    return FABGPU_OK;
}

}  // extern "C"
