// +build fabgpu

package gpu

// Test helper: a block of endorser transactions built with the reference's OWN client-side helpers (protoutil.CreateChaincodeProposal,
// CreateProposalResponse, CreateSignedTx - protoutil/proputils.go:23-119, protoutil/txutils.go:134-298), signed by throw-away P-256
// identities whose certificates are minted here.  Returns the marshalled block and, per signature of the block, the
// (imported key, signed message, DER signature) triple the validators will present to bccsp.Verify.

import (
	"crypto/ecdsa"
	"crypto/elliptic"
	"crypto/rand"
	"crypto/sha256"
	"crypto/x509"
	"crypto/x509/pkix"
	"encoding/pem"
	"math/big"
	"testing"
	"time"

	"github.com/golang/protobuf/proto"
	"github.com/hyperledger/fabric-protos-go/common"
	mspproto "github.com/hyperledger/fabric-protos-go/msp"
	"github.com/hyperledger/fabric-protos-go/peer"
	"github.com/hyperledger/fabric/bccsp"
	"github.com/hyperledger/fabric/bccsp/utils"
	"github.com/hyperledger/fabric/protoutil"
	"github.com/stretchr/testify/require"
)

type blockTuple struct {
	Key bccsp.Key
	Msg []byte
	Sig []byte
}

// testSigner is a protoutil.Signer over a fresh P-256 key with a self-signed certificate, serialized the way msp identities are
// (msp/identities.go:199-214: SerializedIdentity{Mspid, IdBytes = PEM certificate}); signatures are low-S DER as bccsp/sw makes them.
type testSigner struct {
	priv  *ecdsa.PrivateKey
	ident []byte
	key   bccsp.Key
}

func newTestSigner(t testing.TB, g bccsp.BCCSP, mspID string, serial int64) *testSigner {
	priv, err := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
	require.NoError(t, err)
	tmpl := &x509.Certificate{SerialNumber: big.NewInt(serial), Subject: pkix.Name{CommonName: "fabgpu-test"},
		NotBefore: time.Now().Add(-time.Hour), NotAfter: time.Now().Add(time.Hour)}
	der, err := x509.CreateCertificate(rand.Reader, tmpl, tmpl, &priv.PublicKey, priv)
	require.NoError(t, err)
	cert, err := x509.ParseCertificate(der)
	require.NoError(t, err)
	k, err := g.KeyImport(cert, &bccsp.X509PublicKeyImportOpts{Temporary: true})
	require.NoError(t, err)
	ident := protoutil.MarshalOrPanic(&mspproto.SerializedIdentity{Mspid: mspID, IdBytes: pem.EncodeToMemory(&pem.Block{Type: "CERTIFICATE", Bytes: der})})
	return &testSigner{priv: priv, ident: ident, key: k}
}

func (s *testSigner) Serialize() ([]byte, error) { return s.ident, nil }

func (s *testSigner) Sign(msg []byte) ([]byte, error) {
	d := sha256.Sum256(msg)
	r, sv, err := ecdsa.Sign(rand.Reader, s.priv, d[:])
	if err != nil {
		return nil, err
	}
	sv, _ = utils.ToLowS(&s.priv.PublicKey, sv)
	return utils.MarshalECDSASignature(r, sv)
}

func buildSignedBlock(t testing.TB, g bccsp.BCCSP, nTx, nEndorsers int) ([]byte, []blockTuple) {
	creator := newTestSigner(t, g, "Org1MSP", 1)
	endorsers := make([]*testSigner, nEndorsers)
	for i := range endorsers {
		endorsers[i] = newTestSigner(t, g, "Org1MSP", int64(100+i))
	}
	block := protoutil.NewBlock(7, []byte("previous"))
	var tuples []blockTuple
	for i := 0; i < nTx; i++ {
		cis := &peer.ChaincodeInvocationSpec{ChaincodeSpec: &peer.ChaincodeSpec{ChaincodeId: &peer.ChaincodeID{Name: "cc"},
			Input: &peer.ChaincodeInput{Args: [][]byte{[]byte("invoke"), {byte(i)}}}}}
		prop, _, err := protoutil.CreateChaincodeProposal(common.HeaderType_ENDORSER_TRANSACTION, "testchannel", cis, creator.ident)
		require.NoError(t, err)
		var resps []*peer.ProposalResponse
		for _, e := range endorsers {
			r, err := protoutil.CreateProposalResponse(prop.Header, prop.Payload, &peer.Response{Status: 200}, []byte("rwset"), nil, &peer.ChaincodeID{Name: "cc"}, e)
			require.NoError(t, err)
			resps = append(resps, r)
			tuples = append(tuples, blockTuple{Key: e.key, Msg: append(append([]byte{}, r.Payload...), r.Endorsement.Endorser...), Sig: r.Endorsement.Signature})
		}
		env, err := protoutil.CreateSignedTx(prop, creator, resps...)
		require.NoError(t, err)
		tuples = append(tuples, blockTuple{Key: creator.key, Msg: env.Payload, Sig: env.Signature})
		block.Data.Data = append(block.Data.Data, protoutil.MarshalOrPanic(env))
	}
	block.Header.DataHash = protoutil.BlockDataHash(block.Data)
	raw, err := proto.Marshal(block)
	require.NoError(t, err)
	return raw, tuples
}
