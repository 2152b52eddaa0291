// Process configuration of libsvhip.so (svh_init / svh_config in include/svh.h): what the engines read instead of the
// environment.  Nothing here is part of the public C-ABI.
#ifndef SVH_CONFIG_H
#define SVH_CONFIG_H

#include "../../include/svh.h"

namespace svh {

// getenv(name) while the SVH_* switches are honoured (the default; svh_config::read_env), NULL otherwise.  Every
// switch of the library goes through this, and every caller evaluates it lazily (function-local statics), i.e. after
// the configuration is fixed.
const char* env(const char* name);

// The implicit svh_init(NULL) of a process that never calls svh_init: the first entry that is about to touch the HIP
// runtime or to start workers calls this (svh_device_count, svh_set_device and the *_create entries).  One relaxed
// load once the configuration is fixed.
void ensure_init();

// An engine's hook: called once with the effective configuration when it is fixed, and again on every later
// svh_init (which may still change workers / pairs per launch / stage / wait).  explicit_call = 0 for the implicit
// initialisation of a first use: the hook then applies the environment switches only and leaves alone what the
// program may already have set through svh_elas_set_*.  Registered from a static constructor of the engine's
// translation unit (memory only: loading the library reads and writes nothing outside itself).
void on_config(void (*fn)(const svh_config& effective, int explicit_call));

}   // namespace svh

#endif
