// tn_build_id (include/tinyntt.h): identifies the sources this library was compiled from.  Its own translation unit so that
// the Makefile can recompile it whenever kernels.hip, capi.cpp or a header changes (TN_BUILD_ID = sha256 of all of them).
#ifndef TN_BUILD_ID
#define TN_BUILD_ID "unknown"
#endif
extern "C" __attribute__((visibility("default"))) const char* tn_build_id(void) { return TN_BUILD_ID; }
