"""Race screen of the counted-vmcnt GEMM kernels as a GPU test (tools/race_screen.py, fewer repetitions): the NT kernels must be
bit-identical from run to run, the atomic TN kernels within 2e-5 of fp64."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_counted_vmcnt_kernels_are_deterministic():
  path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'race_screen.py')
  spec = importlib.util.spec_from_file_location('race_screen', path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  assert mod.run(REPS=6, verbose=True) == 0
