"""Host tests for two round-1 advisor findings: (1) an unseeded stepper / deck must not silently run with seed 0
(the reference leaves numpy entropy-seeded and pickles the start state, multi_bodies/multi_bodies.py:1154-1161);
(2) deck options that select other physics (radii_*, *_free_surface, *_no_wall against `domain`, python body-body
forces, per-blob radii in a .vertex file) must raise instead of being ignored (multi_bodies.py:207-287,
multi_bodies_functions.py:249-278, :348-356)."""
import os
import pickle

import numpy as np
import pytest

from _oracle_ctx import OracleContext
from test_rollers_host import _write_deck


def test_unseeded_steppers_draw_different_numbers(oracle):
  from rigidmultiblobswall_amd.rollers import RollersIntegrator
  from rigidmultiblobswall_amd.rigid_integrator import RigidIntegrator
  from rigidmultiblobswall_amd import structures as st
  r0 = np.array([[0.0, 0.0, 1.0], [2.0, 0.0, 1.5], [0.0, 2.0, 1.2], [2.0, 2.0, 2.0]])
  a = [RollersIntegrator(r0, "stochastic_adams_bashforth", 0.4, 1.0, device="cpu", ctx=OracleContext(oracle)) for _ in range(2)]
  assert a[0].seed != a[1].seed
  assert not np.array_equal(a[0]._randn(12).numpy(), a[1]._randn(12).numpy())
  # an explicit seed stays reproducible
  b = [RollersIntegrator(r0, "stochastic_adams_bashforth", 0.4, 1.0, device="cpu", ctx=OracleContext(oracle), seed=5) for _ in range(2)]
  assert np.array_equal(b[0]._randn(12).numpy(), b[1]._randn(12).numpy())
  shell = st.icosahedron_shell(0.5)
  loc = np.array([[0.0, 0.0, 2.0], [3.0, 0.0, 2.0]])
  quat = np.tile([1.0, 0, 0, 0], (2, 1))
  c = [RigidIntegrator([shell] * 2, loc, quat, "stochastic_first_order_RFD", 0.2, 1.0, device="cpu", ctx=OracleContext(oracle))
       for _ in range(2)]
  assert c[0].seed != c[1].seed
  assert not np.array_equal(c[0]._normal(12).numpy(), c[1]._normal(12).numpy())


def test_unseeded_deck_is_entropy_seeded_and_always_saves_its_state(oracle, tmp_path):
  from rigidmultiblobswall_amd.read_input import ReadInput
  r0 = np.array([[0.0, 0.0, 1.0], [3.0, 0.0, 1.5], [0.0, 3.0, 1.2]])
  deck = _write_deck(tmp_path, r0)
  text = open(deck).read().replace("seed                                   7\n", "")
  open(deck, "w").write(text)
  read = ReadInput(deck)
  assert read.seed is None
  g1, g2 = read.random_generator(save=True), read.random_generator(save=False)
  assert not np.array_equal(g1.randn(8), g2.randn(8))
  state_file = read.output_name + ".random_state"
  assert os.path.exists(state_file)
  # the pickled state reproduces the run's stream
  g3 = np.random.RandomState()
  with open(state_file, "rb") as fh:
    g3.set_state(pickle.load(fh))
  g1b = read.random_generator(save=True)
  with open(state_file, "rb") as fh:
    st = pickle.load(fh)
  g4 = np.random.RandomState(); g4.set_state(st)
  assert np.array_equal(g1b.randn(8), g4.randn(8))


@pytest.mark.parametrize("edit,needle", [
    (("mobility_vector_prod_implementation    pycuda", "mobility_vector_prod_implementation    radii_numba"), "radii"),
    (("mobility_vector_prod_implementation    pycuda", "mobility_vector_prod_implementation    numba_free_surface"), "free surface"),
    (("mobility_vector_prod_implementation    pycuda", "mobility_vector_prod_implementation    pycuda_no_wall"), "domain"),
    (("mobility_vector_prod_implementation    pycuda", "mobility_vector_prod_implementation    pycuda\ndomain no_wall"), "domain"),
    (("blob_blob_force_implementation         pycuda", "blob_blob_force_implementation         radii_numba"), "radii"),
    (("blob_blob_force_implementation         pycuda", "blob_blob_force_implementation         pycuda\nbody_body_force_torque_implementation python"), "body-body"),
    (("mobility_vector_prod_implementation    pycuda", "mobility_vector_prod_implementation    fortran"), "unknown"),
])
def test_roller_decks_with_modes_not_built_raise(oracle, tmp_path, edit, needle):
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import rollers
  r0 = np.array([[0.0, 0.0, 1.0], [3.0, 0.0, 1.5], [0.0, 3.0, 1.2]])
  deck = _write_deck(tmp_path, r0)
  text = open(deck).read()
  assert edit[0] in text
  open(deck, "w").write(text.replace(edit[0], edit[1]))
  with pytest.raises(ValueError, match=needle):
    rollers.integrator_from_input(ReadInput(deck), device="cpu", ctx=OracleContext(oracle))


def test_consistent_no_wall_deck_is_accepted(oracle, tmp_path):
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import rollers
  r0 = np.array([[0.0, 0.0, 1.0], [3.0, 0.0, 1.5], [0.0, 3.0, 1.2]])
  deck = _write_deck(tmp_path, r0, extra="domain no_wall")
  text = open(deck).read().replace("mobility_vector_prod_implementation    pycuda", "mobility_vector_prod_implementation    numba_no_wall")
  open(deck, "w").write(text)
  integ = rollers.integrator_from_input(ReadInput(deck), device="cpu", ctx=OracleContext(oracle))
  assert integ.domain == "no_wall"


def _rigid_deck(tmp_path, vertex_text, extra=""):
  (tmp_path / "body.vertex").write_text(vertex_text)
  (tmp_path / "body.clones").write_text("1\n0 0 3 1 0 0 0\n")
  deck = tmp_path / "inputfile.dat"
  deck.write_text("""scheme deterministic_forward_euler
mobility_blobs_implementation python
mobility_vector_prod_implementation numba
blob_radius 0.25
eta 1.0
dt 0.01
n_steps 1
output_name %s
structure body.vertex body.clones
%s
""" % (str(tmp_path / "run"), extra))
  return str(deck)


def test_rigid_decks_per_blob_radii_and_dense_block_suffixes(oracle, tmp_path):
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import rigid_integrator as ri
  ok = "3\n0 0 0 0.25\n1 0 0 0.25\n0 1 0 0.25\n"
  integ = ri.integrator_from_input(ReadInput(_rigid_deck(tmp_path, ok)), device="cpu", ctx=OracleContext(oracle))
  assert integ.Nblobs == 3
  bad = "3\n0 0 0 0.25\n1 0 0 0.30\n0 1 0 0.25\n"
  with pytest.raises(ValueError, match="per-blob radii"):
    ri.integrator_from_input(ReadInput(_rigid_deck(tmp_path, bad)), device="cpu", ctx=OracleContext(oracle))
  deck = _rigid_deck(tmp_path, ok)
  text = open(deck).read().replace("mobility_blobs_implementation python", "mobility_blobs_implementation C++_free_surface")
  open(deck, "w").write(text)
  with pytest.raises(ValueError, match="free surface"):
    ri.integrator_from_input(ReadInput(deck), device="cpu", ctx=OracleContext(oracle))
  deck = _rigid_deck(tmp_path, ok)
  text = open(deck).read().replace("mobility_blobs_implementation python", "mobility_blobs_implementation python_no_wall")
  open(deck, "w").write(text)
  with pytest.raises(ValueError, match="domain"):
    ri.integrator_from_input(ReadInput(deck), device="cpu", ctx=OracleContext(oracle))
