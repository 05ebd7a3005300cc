/* Plain-C user of the C ABI (include/fabgpu.h): what a cgo / JNI / ctypes binding sees.  Builds with
 *     gcc -std=c99 -Wall -Iinclude tools/abi_smoke.c -Lfabric-mod_amd/lib -lfabgpu -o abi_smoke
 * and, on an MI355X, verifies one NIST P-256 / SHA-256 signature (RFC 6979 A.2.5, message "sample", with s replaced by its
 * low-S mirror n - s, the form bccsp/sw signs and accepts) through the fused entry point and prints the verdict.
 * tests/test_host_logic.py compiles this file to prove that the headers are valid C and that the symbols link. */
#include <stdio.h>
#include <string.h>

#include "fabgpu.h"
#include "fabgpu_bccsp.h"

static void hex32(const char* h, uint8_t* out) {
    for (int i = 0; i < 32; i++) {
        unsigned v;
        sscanf(h + 2 * i, "%2x", &v);
        out[i] = (uint8_t)v;
    }
}

int main(void) {
    uint8_t qx[32], qy[32], r[32], s[32];
    hex32("60FED4BA255A9D31C961EB74C6356D68C049B8923B61FA6CE669622E60F29FB6", qx);
    hex32("7903FE1008B8BC99A41AE9E95628BC64F2F1B20C2D7E9F5177A3C294D4462299", qy);
    hex32("EFD48B2AACB6A8FD1140DD9CD45E81D69D2C877B56AAF991C34D0EA84EAF3716", r);
    hex32("0834E36AD29A83BF2BC9385E491D6099C8FDF9D1ED67AA7EA5F51F93782857A9", s);   /* n - F7CB1C94... (low-S) */
    const uint8_t msg[] = "sample";
    const uint32_t off[2] = {0, 6};
    fabgpu_ctx* ctx = NULL;
    int rc = fabgpu_init(NULL, &ctx);
    if (rc != FABGPU_OK) {
        printf("fabgpu_init: %s (the caller would keep using bccsp/sw)\n", fabgpu_strerror(rc));
        return rc == FABGPU_ENODEV ? 0 : 1;
    }
    uint64_t bits = 0;
    uint8_t status = 0xFF;
    rc = fabgpu_sha256_p256_verify_batch(ctx, 1, msg, off, qx, qy, r, s, &bits, &status);
    printf("rc=%d verdict=%d status=%d (abi %d, %d registered keys)\n", rc, (int)(bits & 1), (int)status, fabgpu_abi_version(), fabgpu_p256_key_count(ctx));
    fabgpu_shutdown(ctx);
    return !(rc == FABGPU_OK && (bits & 1) == 1 && status == FABGPU_ST_VALID);
}
