// Library-level pieces of the C ABI: error text and version.
#include "egoprior_host.h"

namespace ep_host {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace ep_host

extern "C" const char* egoprior_last_error(void) { return ep_host::error_buffer(); }
extern "C" int egoprior_abi_version(void) { return EGOPRIOR_ABI_VERSION; }
