"""rigidmultiblobswall_amd/utilities.py (schemes of multi_bodies/multi_bodies_utilities.py) on CPU tensors with the
oracle-backed context, against the files the reference's own script wrote for the same decks (golden g10;
g10_config1_body_mobility is BASELINE.json configs[0], the reference's inputfile_body_mobility.dat)."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err
from _oracle_ctx import OracleContext
from _rigid_common import write_case

CASES = golden_files("g10_*.npz")


def _run(g, tmp_path, device, ctx):
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import utilities
  deck = write_case(g, str(tmp_path))
  if "force_file" in g:
    np.savetxt(os.path.join(str(tmp_path), "force.dat"), g["force_file"])
    np.savetxt(os.path.join(str(tmp_path), "velocity.dat"), g["velocity_file"])
  # write_case rewrites `output_name ... run`; the config-1 deck uses its own output name
  text = open(deck).read().replace("output_name                              run.body_mobility",
                                   "output_name                              " + os.path.join(str(tmp_path), "run.body_mobility"))
  open(deck, "w").write(text)
  read = ReadInput(deck)
  return read, utilities.run(read, device=device, ctx=ctx)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[4:-4] for p in CASES])
def test_utilities_scheme_matches_reference_script(oracle, tmp_path, path):
  g = load_golden(path)
  read, out = _run(g, tmp_path, "cpu", OracleContext(oracle))
  for key in ("velocity", "force", "body_mobility", "body_slip_mobility"):
    if key in g:
      written = np.loadtxt(read.output_name + "." + key + ".dat")
      assert rel_err(written, g[key]) < 1e-9, (key, rel_err(written, g[key]))
      assert rel_err(out[key], g[key]) < 1e-9


def test_config1_value():
  g = load_golden(golden_files("g10_config1_body_mobility.npz")[0])
  assert abs(g["body_mobility"][0, 0] - 0.1507443074534127) < 1e-15


def test_unknown_scheme_is_refused(oracle, tmp_path):
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import utilities
  g = load_golden(golden_files("g10_util_resistance.npz")[0])
  deck = write_case(g, str(tmp_path))
  text = open(deck).read().replace("resistance", "plot_velocity_field")
  open(deck, "w").write(text)
  with pytest.raises(ValueError):
    utilities.run(ReadInput(deck), device="cpu", ctx=OracleContext(oracle))
