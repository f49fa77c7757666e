#!/usr/bin/env python3
"""generator of myriad_amd/csrc/dbg_regfill.h (the asm text of the register filler behind MYRIAD_REG_FILL)"""
import os
lines = [f"v_mov_b32 v{i}, %0" for i in range(1,256)] + [f"v_accvgpr_write_b32 a{i}, %0" for i in range(256)]
print("\\n\\t".join(lines))
