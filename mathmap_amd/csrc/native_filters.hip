// placeholder, replaced below
#include "native_filters.h"
namespace mm {
void *NativeWorkspace::reserve(size_t bytes) {
    if (bytes > scratch_bytes) {
        if (scratch) (void)hipFree(scratch);
        scratch = nullptr;
        if (hipMalloc(&scratch, bytes) != hipSuccess) { scratch_bytes = 0; return nullptr; }
        scratch_bytes = bytes;
    }
    return scratch;
}
void NativeWorkspace::release() { if (scratch) (void)hipFree(scratch); scratch = nullptr; scratch_bytes = 0; }
int run_native_filter(const std::string &func, const HNativeRec &, const std::vector<HImageDesc> &, int, int, float *,
                      NativeWorkspace &, hipStream_t, std::string *err) {
    *err = "native filter " + func + " is not implemented yet";
    return -1;
}
}
