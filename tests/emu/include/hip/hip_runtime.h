// TEST INFRASTRUCTURE ONLY: what `#include <hip/hip_runtime.h>` resolves to in the CPU emulation build (build.py --emu).
#pragma once
#include "../../hip_emu.h"
