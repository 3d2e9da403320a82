// Inside librbt.so only (not part of include/rbt.h).
#pragma once
#include "../../include/rbt.h"
extern "C" {
// rbt_last_error describes the LAST call; a function built on other entry points (the V3C walk) ends by putting back the text of the call that failed
__attribute__((visibility("hidden"))) void rbt_internal_set_error(rbt_ctx* ctx, const char* text);
}
