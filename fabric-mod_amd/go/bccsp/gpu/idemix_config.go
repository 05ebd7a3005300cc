/*
SPDX-License-Identifier: Apache-2.0
*/

package gpu

import (
	"github.com/golang/protobuf/proto"
	cb "github.com/hyperledger/fabric-protos-go/common"
	mspprotos "github.com/hyperledger/fabric-protos-go/msp"
)

// the reference's names for these (common/channelconfig/{channel,organization}.go: ApplicationGroupKey, OrdererGroupKey, MSPKey;
// msp/msp.go: IDEMIX ProviderType = 1) - restated so that bccsp/gpu does not import common/channelconfig, which imports bccsp
const (
	applicationGroupKey = "Application"
	ordererGroupKey     = "Orderer"
	mspValueKey         = "MSP"
	idemixProviderType  = 1
)

// RegisterIdemixMSPsOfConfig announces every idemix MSP of a channel's COMMITTED configuration to the block pass, under the channel's
// name (RegisterIdemixMSP: a channel's latest key for an MSP id replaces its earlier one).  Called from the callback that installs the
// channel's MSP manager when a configuration is committed - core/peer/peer.go:295-298 mspCallback, which has cid and the bundle:
//
//	if p, ok := factory.GetDefault().(*gpu.Provider); ok {
//	    p.RegisterIdemixMSPsOfConfig(cid, bundle.ConfigtxValidator().ConfigProto())
//	}
//
// - not from msp/idemixmsp.go Setup, which also runs for configurations that are only being validated and does not know its channel.
// Returns how many MSPs the device accepted.
func (p *Provider) RegisterIdemixMSPsOfConfig(channelID string, config *cb.Config) int {
	if config == nil || config.ChannelGroup == nil {
		return 0
	}
	n := 0
	for _, key := range []string{applicationGroupKey, ordererGroupKey} {
		group, ok := config.ChannelGroup.Groups[key]
		if !ok || group == nil {
			continue
		}
		for _, org := range group.Groups {
			if org == nil {
				continue
			}
			value, ok := org.Values[mspValueKey]
			if !ok || value == nil {
				continue
			}
			mspConfig := &mspprotos.MSPConfig{}
			if proto.Unmarshal(value.Value, mspConfig) != nil || mspConfig.Type != idemixProviderType {
				continue
			}
			idemixConfig := &mspprotos.IdemixMSPConfig{}
			if proto.Unmarshal(mspConfig.Config, idemixConfig) != nil {
				continue
			}
			if p.RegisterIdemixMSP(channelID, idemixConfig.Name, idemixConfig.Ipk) {
				n++
			}
		}
	}
	return n
}
