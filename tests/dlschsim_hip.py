#!/usr/bin/env python3
"""nr_dlschsim-shaped harness for libldpc_hip.so (test infrastructure): the reference's PDSCH transport-channel
acceptance run (openair1/SIMULATION/NR_PHY/dlschsim.c:102-683) with its option letters and result lines
(Tbslbrm = 950984 as there, :140; `PDSCH test OK` when BLER < 0.01, :598-601).

  python tests/dlschsim_hip.py -R 106 -m9 -s13 -n100                 # CI test 1 (test_case_list.xml:232-246): PDSCH test OK
  python tests/dlschsim_hip.py -R 273 -m19 -s20 -n100                # CI test 3
  python tests/dlschsim_hip.py -m15 -s5 -n20 --backend segment        # per-segment LDPCencoder / LDPCdecoder like the sim
  python tests/dlschsim_hip.py -m15 -s5 -n20 --oracle                 # same seeds through the CPU oracle

Flow, back ends and the deliberate differences from the reference sim: tests/nr_schsim.py.
"""
import sys

import nr_schsim

if __name__ == "__main__":
    res = nr_schsim.run("dl", nr_schsim.parser("dl").parse_args())
    sys.exit(min(res["n_errors"], 255))                         # dlschsim.c:682 return (n_errors)
