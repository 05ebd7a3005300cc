// C++ host mirror of the reference's idemix BCCSP verbs on the creator-signature path:
//   bccsp/idemix/handlers/issuer.go   IssuerPublicKeyImporter.KeyImport   -> IdemixCSP::IssuerKeyImport
//   bccsp/idemix/handlers/nym.go:145-167  NymPublicKeyImporter.KeyImport  -> IdemixCSP::NymKeyImport
//   bccsp/idemix/handlers/nymsigner.go:62-95  NymVerifier.Verify          -> IdemixCSP::NymVerifyBatch
//   (what msp/idemixmsp.go:584-599 idemixidentity.Verify calls per creator signature)
// Same argument meaning and error text; every cryptographic verdict comes from the GPU through include/fabgpu.h
// (fabgpu_idemix_nym_verify_batch).  Tuples the device does not decide are flagged needs_sw: the Go provider hands those -
// and everything else of the idemix BCCSP (credentials, revocation, Signature.Ver with its pairings) - to bccsp/idemix.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "bccsp_host.h"

namespace fab {
namespace bccsp {

struct IdemixIssuerPublicKey {      // handlers.issuerPublicKey: what the nym verifier needs of idemix.IssuerPublicKey
    uint8_t hsk_x[32], hsk_y[32], hrand_x[32], hrand_y[32], hash[32];
    int64_t issuer_id = -1;         // device registration (fabgpu_idemix_issuer_register); -1: not accelerated
};
struct NymPublicKey {               // handlers.nymPublicKey: Ecp from bridge.User.NewPublicNymFromBytes
    uint8_t x[32], y[32];
    bool halves_are_32 = false;     // raw was 64 bytes (anything else is for bccsp/sw: amcl's FromBytes on other lengths)
};
struct NymVerifyItem {
    const NymPublicKey* key;            // nullptr: "invalid key, expected *nymPublicKey"
    const IdemixIssuerPublicKey* ipk;   // nullptr: "invalid options, missing issuer public key"
    const uint8_t* sig;                 // marshalled idemix.NymSignature
    size_t siglen;
    const uint8_t* digest;              // the message (idemix signs the message itself: msp/idemixmsp.go:590-593)
    size_t dlen;
};

class IdemixCSP {
   public:
    explicit IdemixCSP(fabgpu_ctx* ctx) : ctx_(ctx) {}
    // raw: marshalled idemix.IssuerPublicKey (idemix/idemix.proto).  Extracts HSk, HRand, Hash and registers the issuer on the
    // device.  The proof of knowledge inside the key (IssuerPublicKey.Check, idemix/issuerkey.go:114-172: G2 arithmetic) is NOT
    // re-checked here - key import stays with bccsp/idemix, which calls this after its own checks passed.
    Error IssuerKeyImport(const uint8_t* raw, size_t len, IdemixIssuerPublicKey& out) const;
    // the fields alone (HSk, HRand, Hash as 32-byte halves), no device: false when the bytes do not parse or a field has another size
    static bool IssuerKeyFields(const uint8_t* raw, size_t len, IdemixIssuerPublicKey& out);
    // Would golang/protobuf re-marshal the key it unmarshals from these bytes to the SAME bytes?  Only then does the library's issuer hash
    // (the bytes minus field 10) equal SetHash's (idemix/issuerkey.go:171-182: the re-marshalled key with Hash cleared); a key in any
    // other encoding is not accelerated (issuer_id -1).
    static bool IssuerKeyEncodingIsCanonical(const uint8_t* raw, size_t len);
    // raw: x || y (bccsp/idemix/bridge/user.go:72-86 splits at len/2)
    Error NymKeyImport(const uint8_t* raw, size_t len, NymPublicKey& out) const;
    Error NymVerifyBatch(const std::vector<NymVerifyItem>& items, std::vector<VerifyResult>& results) const;

   private:
    fabgpu_ctx* ctx_;
};

// idemix.NymSignature{proof_c = 1, proof_s_sk = 2, proof_s_r_nym = 3, nonce = 4}: false when the bytes are not a protobuf message
struct NymSignatureFields {
    const uint8_t* f[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t len[4] = {0, 0, 0, 0};
};
bool UnmarshalNymSignature(const uint8_t* raw, size_t len, NymSignatureFields& out);

}  // namespace bccsp
}  // namespace fab
