#!/usr/bin/env python3
"""nr_ulschsim-shaped harness for libldpc_hip.so (test infrastructure): the reference's PUSCH transport-channel
acceptance run (openair1/SIMULATION/NR_PHY/ulschsim.c:124-647) with its option letters and result lines.

  python tests/ulschsim_hip.py -R 106 -m9 -s13 -n100                 # CI test 1 (test_case_list.xml:263-277): PUSCH test OK
  python tests/ulschsim_hip.py -R 106 -m9 -s13 -n100 -y4 -z4 -W4     # CI test 4: four layers
  python tests/ulschsim_hip.py -m9 -s1 -n20 --backend segment         # per-segment LDPCencoder / LDPCdecoder like the sim
  python tests/ulschsim_hip.py -m9 -s1 -n20 --oracle                  # same seeds through the CPU oracle

Flow, back ends and the deliberate differences from the reference sim: tests/nr_schsim.py.
"""
import sys

import nr_schsim

if __name__ == "__main__":
    res = nr_schsim.run("ul", nr_schsim.parser("ul").parse_args())
    sys.exit(min(res["n_errors"], 255))                         # ulschsim.c:646 return (n_errors)
