// Host side of the aggregate-verify engine: contexts, workspaces, the stage pipeline and the C ABI declared in
// include/bgls_hip.h.  Pipeline of bgls.VerifyAggregateSignature (bgls/bgls.go:94-119) on the device:
//   dup_check     exact duplicate-message scan          (containsDuplicateMessage, bgls.go:139-150)
//   hash_to_g1    H(m_i) for every message              (concurrentHash, bgls.go:107-111,134-137)
//   g1_parse      -sigma                                (aggsig.Mul(-1), bgls.go:112)
//   miller        Miller values of every (H(m_i), pk_i) and of (-sigma, g2), multiplied into per-group partial products
//                                                       (concurrentPair + GT Add tree, curves/curve.go:132-169,217-223)
//   reduce        product of the partial products
//   final36       ONE final exponentiation, compare with 1 (Equals(GetGTIdentity), bgls.go:115-118)
// Kernels live in the k_*.hip units and are reached through launch.hpp.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include <atomic>
#include <string>
#include <thread>
#include <memory>
#include <optional>
#include <unordered_map>
#include <dlfcn.h>
// RCCL is reached through dlopen only (see struct Rccl): the few types and constants it needs are declared here, so the
// library also builds where the RCCL development headers are not installed.  Values follow nccl.h's ABI.
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;

#include <new>
#include <random>
#include <chrono>
#include <stdexcept>
#include <system_error>
#include "dev_common.hpp"
#include "coop.hpp"
#include "coop_r28.hpp"
#include "finalexp.hpp"
#include "miller_kernels.hpp"
#include "launch.hpp"
#include "launch_tail.hpp"
#include "../../include/bgls_hip.h"

using namespace bgls;

// The host side is one translation unit in four files (round 5; it was one 2 657-line file):
#include "engine_core.inc"      // errors, exception barrier, contexts, switches, Engine<C>: the stages of a verification
#include "engine_verify.inc"    // per-call bodies: verifications, point operations, HAE / multiplicity, wire formats, signing
#include "engine_keys.inc"      // resident key sets, multi-device shards, RCCL exchange
#include "abi.inc"              // extern "C": include/bgls_hip.h
