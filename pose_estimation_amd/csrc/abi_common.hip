// ABI version and thread-local error text.
#include "common.h"

namespace ispa {
char *last_error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace ispa

extern "C" int ispa_abi_version(void) { return ISPA_ABI_VERSION; }
extern "C" const char *ispa_last_error(void) { return ispa::last_error_buffer(); }
