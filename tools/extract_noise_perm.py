#!/usr/bin/env python3
"""Regenerates include/pt_noise_perm.h (the Perlin permutation DATA table) from the reference's core/texture.rs."""
import re, sys
src = open(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/core/texture.rs").read()
a = src.index("const NOISE_PERM: [usize; 2 * NOISE_PERM_SIZE] = ["); b = src.index("];", a)
body = re.sub(r"//.*", "", src[a:b][src[a:b].index("= [") + 3:])
nums = [int(x) for x in re.findall(r"\d+", body)]
assert len(nums) == 512 and nums[:256] == nums[256:]
lines = ["    " + ", ".join(str(v) for v in nums[i:i + 32]) + "," for i in range(0, 512, 32)]
print("/* Perlin noise permutation table: DATA from core/texture.rs:26-67 (NOISE_PERM, 2 x 256 entries), extracted by\n"
      " * tools/extract_noise_perm.py. Shared by the oracle and the HIP path like the Sobol' matrices. */\n#pragma once\n#include <stdint.h>\n#define PT_NOISE_PERM_VALUES \\\n"
      + " \\\n".join(lines).rstrip(","))
