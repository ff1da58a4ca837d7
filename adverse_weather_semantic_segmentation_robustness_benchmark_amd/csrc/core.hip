// core.hip — ABI bookkeeping entry points of libawseg_hip.so.
#include "awseg_common.h"

AWSEG_API int awseg_abi_version(void) { return 1; }

#ifndef AWSEG_HEADER_HASH
#define AWSEG_HEADER_HASH 0ULL
#endif
AWSEG_API unsigned long long awseg_header_hash(void) { return AWSEG_HEADER_HASH; }     // csrc/build.py: 60 bits of sha256(include/awseg.h)

AWSEG_API const char* awseg_error_string(int code)
{
    switch (code) {
        case 0: return "success";
        case AWSEG_EINVAL: return "awseg: invalid argument (null pointer, bad size or unknown enum)";
        case AWSEG_ERANGE: return "awseg: size exceeds kernel indexing range";
        case AWSEG_EALIGN: return "awseg: pointer not aligned as documented";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "awseg: unknown error";
}

AWSEG_API int awseg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
