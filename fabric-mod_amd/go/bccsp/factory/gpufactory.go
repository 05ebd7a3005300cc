// +build fabgpu

// GPUFactory: the bccsp/factory side of ProviderName "GPU" (pattern: bccsp/factory/pkcs11factory.go:17-45).
// Drop into the reference tree as bccsp/factory/gpufactory.go and apply the two `case`s of gpufactory_patch.txt to
// nopkcs11.go / pkcs11.go (initFactories and GetBCCSPFromOpts switch on the provider name; there is no runtime BCCSP plugin
// loader in this version - SURVEY.md 8(b) "Selection").  core.yaml:  peer.BCCSP.Default: GPU  (+ the usual SW section, which
// configures the embedded software provider: hash family, security level, key store).
// NOT compiled in this repository (no Go toolchain in the build image); Go 1.14 compatible.
package factory

import (
	"github.com/hyperledger/fabric/bccsp"
	"github.com/hyperledger/fabric/bccsp/gpu"
	"github.com/pkg/errors"
)

const (
	// GPUBasedFactoryName is the name of the factory of the MI355X-accelerated BCCSP implementation
	GPUBasedFactoryName = "GPU"
)

// GPUOpts is the `GPU:` section of the BCCSP configuration.
type GPUOpts struct {
	// Devices: the HIP ordinals this node's ONE provider drives, one device context each (an ordinal may repeat: several contexts on
	// one GPU).  Empty = every visible device.  factory.GetDefault() is process-global (bccsp/factory/factory.go:41-55) and every
	// channel's validator receives it (core/peer/peer.go:337-355), so "one provider per device" cannot be configured - one provider
	// owns them all and routes each block pass to the least busy device (fabgpu_csp_new2, include/fabgpu_bccsp.h).
	Devices []int `mapstructure:"devices" json:"devices" yaml:"Devices"`
	// Device: DEPRECATED alias from the time a provider drove one device (`GPU: Device: N`).  mapstructure ignores unknown keys, so
	// without this field an old core.yaml would silently mean "every visible device" - contexts and ConcurrentPasses' pinned memory
	// on all of them.  Used only when Devices is empty: Devices = [Device].  Setting both to different things is an error.
	Device *int `mapstructure:"device" json:"device,omitempty" yaml:"Device,omitempty"`
	// ConcurrentPasses: per device, how many overlapping block passes to allocate for when the provider is made (staging slots, pinned
	// memo tables, pass arrays) instead of when passes first overlap.  2 suits the arrival pipeline of one channel per device.
	ConcurrentPasses int `mapstructure:"concurrentpasses" json:"concurrentpasses" yaml:"ConcurrentPasses"`
	// ExpectBlockBytes / ExpectTuples size that pre-allocation; 0 = 64 MiB / 65 536 signatures (configtx.yaml AbsoluteMaxBytes and
	// MaxMessageCount x (1 + endorsements) are the numbers to put here).
	ExpectBlockBytes int `mapstructure:"expectblockbytes" json:"expectblockbytes" yaml:"ExpectBlockBytes"`
	ExpectTuples     int `mapstructure:"expecttuples" json:"expecttuples" yaml:"ExpectTuples"`
	// MemoBlocks: how many pre-verified blocks the verdict memo may hold while they wait for their validators - MemoBlocks x ExpectTuples
	// entries.  The bounded memo drops its OLDEST block first, which during a state transfer is the one the committer needs next, so this
	// must not be smaller than what the payload buffer accepts ahead: peer.gossip.state.blockBufferSize (core.yaml, default 20; the
	// arrival hook only pre-verifies blocks the buffer kept).  0 = the library's 2^18 entries (six 10 000-transaction blocks).
	MemoBlocks int `mapstructure:"memoblocks" json:"memoblocks" yaml:"MemoBlocks"`
	// NoHashMemo switches the digest memo off: bccsp.Hash always goes to bccsp/sw and no pass keeps a host copy of its block.  By default a
	// pass keeps its block in pinned host memory of the library until the block is evicted (HashMemoBlocks copies per device at a time;
	// 0: MemoBlocks + ConcurrentPasses, at least 8 - each as large as its block), and identity.Verify's Hash (msp/identities.go:173-181)
	// is answered from it when the validator's bytes equal the block's, byte for byte.
	NoHashMemo     bool `mapstructure:"nohashmemo" json:"nohashmemo" yaml:"NoHashMemo"`
	HashMemoBlocks int  `mapstructure:"hashmemoblocks" json:"hashmemoblocks" yaml:"HashMemoBlocks"`
	// KeyTables16: every public key the provider imports (the endorsers' and creators' keys of the channels' MSPs - msp/cache/cache.go:14-18
	// keeps a hundred identities) also gets a 16-bit comb table on the device: 80 MiB each, built behind the import, the first 64 keys.
	// Launches whose keys all have one verify with 16 mixed additions for u2*Q instead of 32: registered-key verification 170 -> 214-231 M/s
	// on an MI355X; 5 GiB of its 288 GB.
	KeyTables16 bool `mapstructure:"keytables16" json:"keytables16" yaml:"KeyTables16"`
	// HostWalk keeps the envelope walk of the block pass on the host (A/B runs); PassTiming prints every pass's stage breakdown.
	HostWalk   bool `mapstructure:"hostwalk" json:"hostwalk" yaml:"HostWalk"`
	PassTiming bool `mapstructure:"passtiming" json:"passtiming" yaml:"PassTiming"`
	// CoalesceVerify: Verify calls that miss the verdict memo share device launches with whatever other misses are in flight
	// (fabgpu_csp_verify_coalesced).  For the orderer (General.BCCSP in orderer.yaml): its Broadcast handlers reach
	// identity.Verify one message per goroutine behind SigFilter and no block pass precedes them.  Leave it off on peers.
	CoalesceVerify bool `mapstructure:"coalesceverify" json:"coalesceverify" yaml:"CoalesceVerify"`
}

// GPUFactory is the factory of the GPU-accelerated BCCSP.
type GPUFactory struct{}

// Name returns the name of this factory
func (f *GPUFactory) Name() string {
	return GPUBasedFactoryName
}

// Get returns an instance of BCCSP using Opts: the software provider built from SwOpts exactly as SWFactory builds it,
// wrapped by the GPU provider.  If no usable device exists the error is returned: a peer configured for "GPU" should not
// silently run without one (operators who want the fallback configure "SW").
func (f *GPUFactory) Get(config *FactoryOpts) (bccsp.BCCSP, error) {
	if config == nil || config.SwOpts == nil {
		return nil, errors.New("Invalid config. It must not be nil.")
	}
	swCSP, err := (&SWFactory{}).Get(config)
	if err != nil {
		return nil, errors.Wrapf(err, "Failed initializing the software BCCSP behind the GPU provider")
	}
	var opts gpu.Options
	if g := config.GPUOpts; g != nil {
		if g.Device != nil {
			if len(g.Devices) == 0 {
				g.Devices = []int{*g.Device}
			} else if len(g.Devices) != 1 || g.Devices[0] != *g.Device {
				return nil, errors.Errorf("Invalid GPU opts: the deprecated Device [%d] contradicts Devices %v", *g.Device, g.Devices)
			}
		}
		opts = gpu.Options{Devices: g.Devices, ConcurrentPasses: g.ConcurrentPasses, ExpectBlockBytes: g.ExpectBlockBytes,
			ExpectTuples: g.ExpectTuples, MemoBlocks: g.MemoBlocks, HostWalk: g.HostWalk, PassTiming: g.PassTiming,
			NoHashMemo: g.NoHashMemo, HashMemoBlocks: g.HashMemoBlocks, KeyTables16: g.KeyTables16}
	}
	csp, err := gpu.New(swCSP, opts)
	if err != nil {
		return nil, err
	}
	if config.GPUOpts != nil && config.GPUOpts.CoalesceVerify {
		if p, ok := csp.(*gpu.Provider); ok {
			p.SetCoalesce(true)
		}
	}
	return csp, nil
}
