// Probe (no GPU memory access, CPU-side queries only): what does the HIP runtime believe about addresses NEXT TO a range registered
// with hipHostRegister when the range is not page-aligned?  Written after a GPU memory fault on a page-aligned host-heap address in
// the first host-vector call that read a registered numpy array whose neighbour in the heap was the (pageable) output array.
// build: hipcc -O1 -o host_register_range host_register_range.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static void query(const char* what, const void* p) {
    hipPointerAttribute_t a; memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    void* dp = nullptr;
    hipError_t e2 = hipHostGetDevicePointer(&dp, const_cast<void*>(p), 0);
    printf("%-52s %p : attributes %-22s type=%d host=%p device=%p | hipHostGetDevicePointer %-22s -> %p\n", what, p, hipGetErrorName(e), (int)a.type,
           a.hostPointer, a.devicePointer, hipGetErrorName(e2), dp);
    (void)hipGetLastError();
}
int main() {
    char* buf = nullptr;
    if (posix_memalign((void**)&buf, 4096, 4 * 4096)) return 1;
    memset(buf, 0, 4 * 4096);
    char* reg = buf + 64; const size_t bytes = 4096 + 1000;          // registered: [buf + 64, buf + 5160): ends inside page 1
    printf("buffer %p (4 pages); registering [%p, %p)\n", (void*)buf, (void*)reg, (void*)(reg + bytes));
    hipError_t e = hipHostRegister(reg, bytes, hipHostRegisterDefault);
    printf("hipHostRegister: %s\n", hipGetErrorName(e));
    query("start of the registered range", reg);
    query("inside the registered range", reg + 2000);
    query("last byte of the registered range", reg + bytes - 1);
    query("16 bytes past its end (same page)", reg + bytes + 16);
    query("before its start (same page)", buf + 8);
    query("end of the page the range ends in (page 1, last byte)", buf + 2 * 4096 - 1);
    query("next page (page 2)", buf + 2 * 4096 + 8);
    e = hipHostUnregister(reg);
    printf("hipHostUnregister: %s\n", hipGetErrorName(e));
    query("start of the range after unregistering", reg);
    free(buf);
    return 0;
}
