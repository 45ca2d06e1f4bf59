#!/usr/bin/env python
"""Workload for the PMC passes on the spectral match (test infrastructure): two launches of witw_match_fwd_dft at 16384 x 4096.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p --output-format csv -- python3 tools/pmc_match_dft.py
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d out2 -o p --output-format csv -- python3 tools/pmc_match_dft.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from witw_amd import ops  # noqa: E402

gen = torch.Generator(device='cuda')
gen.manual_seed(1)
ov = torch.randn((16384, 16, 4, 64), generator=gen, device='cuda')
su = torch.randn((4096, 16, 4, 64), generator=gen, device='cuda')
sg, sq = ops.match_spectrum(ov, overhead=True), ops.match_spectrum(su, overhead=False)
for _ in range(2):
    ops.match_fwd_dft(ov, su, spec_ov=sg, spec_su=sq, want_orientation=False)
torch.cuda.synchronize()
print('done', flush=True)
