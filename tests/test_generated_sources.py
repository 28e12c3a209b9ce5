"""Generated sources are committed next to their generators; these checks keep the two in step (no GPU needed)."""
import os
import subprocess
import sys

from conftest import PKG, ROOT


def test_packed_kernel_sweeps_are_what_their_generator_prints():
    csrc = os.path.join(PKG, "csrc")
    out = subprocess.run([sys.executable, os.path.join(csrc, "gen_pk_sweeps.py")], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert out == open(os.path.join(csrc, "pk_sweeps_gen.inc")).read(), "run: python3 gen_pk_sweeps.py > pk_sweeps_gen.inc (in csrc/)"


def test_cell_microbenchmark_is_what_its_generator_prints():
    mb = os.path.join(ROOT, "tools", "microbench")
    for gen, src in (("gen_cell_v3.py", "cell_v3.hip"), ("gen_cell_order.py", "cell_order.hip"), ("gen_valu_mix.py mix", "valu_mix.hip"),
                     ("gen_valu_mix.py nop", "valu_mix_nop.hip")):
        out = subprocess.run([sys.executable] + [os.path.join(mb, gen.split()[0])] + gen.split()[1:], stdout=subprocess.PIPE, text=True,
                             check=True, cwd=mb).stdout
        assert out == open(os.path.join(mb, src)).read(), "%s is stale: python3 %s > %s" % (src, gen, src)
