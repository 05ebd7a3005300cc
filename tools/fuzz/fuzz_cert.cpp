#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <string>
#include "block_prepass.h"
#include "bccsp_host.h"
#include "idemix_host.h"
using namespace fab::bccsp;
int main() {
    FILE* f = fopen("/tmp/cert.pem", "rb");
    std::vector<uint8_t> pem(1 << 16);
    size_t n = fread(pem.data(), 1, pem.size(), f);
    pem.resize(n);
    std::vector<uint8_t> der;
    if (!PemToDer(pem.data(), pem.size(), der)) { printf("base pem does not decode\n"); return 1; }
    std::mt19937_64 rng(11);
    size_t ok = 0, sigok = 0, windowed = 0;
    for (int it = 0; it < 200000; it++) {
        std::vector<uint8_t> d = der;
        int k = 1 + rng() % 6;
        for (int j = 0; j < k; j++) {
            size_t pos = rng() % d.size();
            switch (rng() % 4) {
                case 0: d[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 1: d[pos] = (uint8_t)rng(); break;
                case 2: d[pos] = 0x84; break;                          // long-form lengths
                default: if (d.size() > 8) d.resize(d.size() - rng() % 8); break;
            }
        }
        uint8_t* heap = (uint8_t*)malloc(d.size());
        memcpy(heap, d.data(), d.size());
        uint8_t qx[32], qy[32];
        ok += CertDerToP256(heap, d.size(), qx, qy);
        {   // the walk the device's certificate decoder shares with the host (block_walk_core.h): the key must lie inside the certificate
            const int32_t at = walk::cert_der_p256_key_offset(heap, d.size());
            if (at >= 0 && (size_t)at + 64 > d.size()) { printf("KEY OFFSET OUT OF RANGE\n"); return 1; }
            // ... and the same walk over a WINDOW of the certificate (the device decoder holds the first 3 KiB of a certificate of any
            // length): an exact-size copy of the window - a read past it is ASan's to catch -, and the answer is the full walk's, or
            // "the key lies beyond the window" (-2), never another offset and never "not a P-256 certificate" for one that is
            const size_t avail = d.size() ? rng() % (d.size() + 1) : 0;
            uint8_t* win = (uint8_t*)malloc(avail ? avail : 1);
            memcpy(win, heap, avail);
            const int32_t aw = walk::cert_der_p256_key_offset_window(win, avail, d.size());
            free(win);
            if (aw >= 0 && (aw != at || (size_t)aw + 64 > avail)) { printf("WINDOW WALK: ANOTHER OFFSET\n"); return 1; }
            if (aw == -1 && at >= 0) { printf("WINDOW WALK: REFUSES A CERTIFICATE THE FULL WALK TAKES\n"); return 1; }
            windowed += aw == -2;
        }
        // the ECDSA signature gate on arbitrary bytes (the tail of the certificate holds a real DER signature)
        BigInt R, S;
        size_t off = d.size() > 80 ? d.size() - 72 - rng() % 8 : 0;
        sigok += UnmarshalECDSASignature(heap + off, d.size() - off, R, S).ok();
        {   // the device route's general gate on the same slice: agrees with the host's parser on "unmarshals with r, s > 0"
            uint8_t* sl = (uint8_t*)malloc(d.size() - off ? d.size() - off : 1);
            memcpy(sl, heap + off, d.size() - off);
            uint32_t pr, lr, ps, ls;
            const uint8_t g = walk::gate_sig_general(sl, (uint32_t)(d.size() - off), pr, lr, ps, ls);
            BigInt R3, S3;
            const bool host_ok = d.size() != off && UnmarshalECDSASignature(sl, d.size() - off, R3, S3).ok();
            const bool dev_ok = g == walk::GATE_SUBMIT || g == walk::GATE_HIGH_S || g == walk::GATE_RANGE;
            if (d.size() != off && host_ok != dev_ok) { printf("GENERAL GATE DISAGREES WITH THE HOST PARSER\n"); return 1; }
            free(sl);
        }
        free(heap);
        // PEM layer: mutate the text
        if (it % 8 == 0) {
            std::vector<uint8_t> p = pem;
            for (int j = 0; j < 3; j++) p[rng() % p.size()] = (uint8_t)rng();
            uint8_t* hp = (uint8_t*)malloc(p.size());
            memcpy(hp, p.data(), p.size());
            std::vector<uint8_t> out;
            PemToDer(hp, p.size(), out);
            free(hp);
        }
    }
    // idemix issuer keys (round 5): the canonical-encoding gate and the field extraction on mutants of a reference key, exact-size copies
    size_t canon = 0, fields = 0;
    if (FILE* fi = fopen("/tmp/ipk.bin", "rb")) {
        std::vector<uint8_t> ipk(1 << 16);
        ipk.resize(fread(ipk.data(), 1, ipk.size(), fi));
        fclose(fi);
        if (!IdemixCSP::IssuerKeyEncodingIsCanonical(ipk.data(), ipk.size())) { printf("THE REFERENCE'S OWN ISSUER KEY IS NOT CANONICAL\n"); return 1; }
        for (int it = 0; it < 200000; it++) {
            std::vector<uint8_t> d = ipk;
            const int k = 1 + (int)(rng() % 4);
            for (int j = 0; j < k; j++) {
                const size_t pos = rng() % d.size();
                switch (rng() % 5) {
                    case 0: d[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
                    case 1: d[pos] = (uint8_t)rng(); break;
                    case 2: d[pos] = 0x80; break;                      // padded varints
                    case 3: d.insert(d.begin() + (long)pos, (uint8_t)rng()); break;
                    default: if (d.size() > 8) d.resize(d.size() - rng() % 8); break;
                }
            }
            uint8_t* hp = (uint8_t*)malloc(d.size() ? d.size() : 1);
            memcpy(hp, d.data(), d.size());
            IdemixIssuerPublicKey out;
            const bool c = IdemixCSP::IssuerKeyEncodingIsCanonical(hp, d.size());
            const bool fl = IdemixCSP::IssuerKeyFields(hp, d.size(), out);
            canon += c;
            fields += fl;
            free(hp);
        }
    }
    printf("issuer keys: %zu of 200000 mutants still canonical, %zu still yield HSk / HRand / Hash\n", canon, fields);
    printf("fuzz ok: %zu mutants still gave a P-256 key, %zu signature slices unmarshalled, %zu window walks asked for more bytes\n", ok, sigok, windowed);
}
