#include "common.h"
#include "medimgen_hip.h"
extern "C" int mi_abi_version(void) { return 8; }
