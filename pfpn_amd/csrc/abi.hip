// ABI version + device-code probe for libpfpn_hip.so.
#include "pfpn_common.h"

extern "C" int pfpn_abi_version(void) { return 1; }
