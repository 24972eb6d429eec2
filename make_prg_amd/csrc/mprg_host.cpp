// mprg_host.cpp — the HOST-only entry points of include/mprg.h (names ending in _host that touch no device: the one-pass
// .bin / .gfa encoders and the FASTA parser) as a library of their own, libmprg_host.so, built with plain g++.
// libmprg_hip.so exports the same functions (mprg_api.hip includes the same source), but loading THAT library pulls in the
// HIP runtime — before torch has loaded its own copy, or in a parent process that is about to fork GPU workers, that is
// exactly what must not happen — so the Python host binds these functions from here.
//   g++ -O3 -std=c++17 -fPIC -shared -pthread make_prg_amd/csrc/mprg_host.cpp -o make_prg_amd/_lib/libmprg_host.so -lz
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "host_encoders.inc"
#include "host_batch.inc"
