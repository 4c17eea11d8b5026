"""Row N3: the reference's input-deck format and backend strings."""
import numpy as np
import pytest


def test_read_input_reference_deck_format(tmp_path):
  from rigidmultiblobswall_amd.read_input import ReadInput
  deck = tmp_path / "inputfile_body_mobility.dat"
  # the option set of multi_bodies/inputfile_body_mobility.dat (config 1) plus a few more
  deck.write_text("""# comment line
scheme                                   body_mobility
mobility_blobs_implementation            hip
eta                                      1.0   # trailing comment
blob_radius                              0.25
output_name                              data/run.body_mobility
periodic_length                          10 12.5 0
structure\tStructures/boomerang_N_15.vertex Structures/boomerang_N_15.clones
structure Structures/shell_N_12_Rg_1.vertex Structures/shell_N_12_Rg_1.clones
some_future_option 42
""")
  ri = ReadInput(str(deck))
  assert ri.scheme == "body_mobility" and ri.eta == 1.0 and ri.blob_radius == 0.25
  assert ri.mobility_blobs_implementation == "hip"
  assert ri.mobility_vector_prod_implementation == "python" and ri.solver_tolerance == 1e-8   # defaults
  assert np.array_equal(ri.periodic_length, [10, 12.5, 0])
  assert ri.structures == [["Structures/boomerang_N_15.vertex", "Structures/boomerang_N_15.clones"],
                           ["Structures/shell_N_12_Rg_1.vertex", "Structures/shell_N_12_Rg_1.clones"]]
  assert ri.structures_ID == ["boomerang_N_15", "shell_N_12_Rg_1"] and ri.num_free_bodies == 2
  assert ri.options["some_future_option"] == "42"


def test_dispatch_strings():
  from rigidmultiblobswall_amd import dispatch, forces, mobility
  assert dispatch.set_mobility_vector_prod("hip") is mobility.single_wall_mobility_trans_times_force_hip
  assert dispatch.set_mobility_vector_prod("hip_no_wall") is mobility.no_wall_mobility_trans_times_force_hip
  assert dispatch.set_mobility_vector_prod("pycuda", accept_reference_gpu_names=True) is \
      mobility.single_wall_mobility_trans_times_force_hip
  assert dispatch.set_blob_blob_forces("hip") is forces.calc_blob_blob_forces_hip
  assert dispatch.set_blob_blob_forces("None")(np.zeros((5, 3))).shape == (5, 3)
  with pytest.raises(ValueError):
    dispatch.set_mobility_vector_prod("numba")      # CPU backends are the reference's, not ours
  with pytest.raises(ValueError):
    dispatch.set_mobility_vector_prod("pycuda")
  assert dispatch.set_mobility_vector_prod("hip_free_surface") is mobility.free_surface_mobility_trans_times_force_hip
  assert dispatch.set_mobility_vector_prod("pycuda_free_surface", accept_reference_gpu_names=True) is \
      mobility.free_surface_mobility_trans_times_force_hip
  # blobs of different radii: the partial carries the radii and the source->target function (multi_bodies.py:266-286)

  class Body(object):
    def __init__(self, radii):
      self.blobs_radius = np.asarray(radii)
  fn = dispatch.set_mobility_vector_prod("radii_hip", bodies=[Body([0.1, 0.2]), Body([0.3])])
  assert fn.func is mobility.mobility_radii_trans_times_force
  assert np.array_equal(fn.keywords["radius_blobs"], [0.1, 0.2, 0.3])
  assert fn.keywords["function"] is mobility.single_wall_mobility_trans_times_force_source_target_hip
  fn = dispatch.set_mobility_vector_prod("radii_hip_no_wall", radius_blobs=[0.5, 0.5])
  assert fn.keywords["function"] is mobility.no_wall_mobility_trans_times_force_source_target_hip
  ff = dispatch.set_blob_blob_forces("radii_hip", radius_blobs=[0.1, 0.2])
  assert ff.func is forces.calc_blob_blob_forces_radii_hip and np.array_equal(ff.keywords["radius_blobs"], [0.1, 0.2])
  fn = dispatch.set_mobility_vector_prod("radii_hip_free_surface", radius_blobs=[0.5])
  assert fn.keywords["function"] is mobility.free_surface_mobility_trans_times_force_source_target_hip
  with pytest.raises(ValueError):
    dispatch.set_mobility_vector_prod("radii_hip")
