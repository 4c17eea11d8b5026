"""Shared by the CPU (oracle-backed) and GPU rigid-integrator tests: replay a g9 fixture.  The fixture holds the deck
the reference's driver was run on; it is written back to disk together with the structure files and handed to
ReadInput -> integrator_from_input -> run, i.e. the test exercises the same entry path a user of the reference has."""
import os

import numpy as np


def write_case(g, tmp_path):
  for ID in [str(x) for x in g["IDs"]]:
    with open(os.path.join(tmp_path, ID + ".vertex"), "w") as fh:
      v = g["vertex_" + ID]
      fh.write("%d\n" % len(v))
      for x in v:
        fh.write("%.17g %.17g %.17g\n" % tuple(x))
    with open(os.path.join(tmp_path, ID + ".clones"), "w") as fh:
      loc, quat = g["locations_" + ID], g["quaternions_" + ID]
      fh.write("%d\n" % len(loc))
      for x, q in zip(loc, quat):
        fh.write("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n" % (tuple(x) + tuple(q)))
    if "slip_" + ID in g:
      with open(os.path.join(tmp_path, ID + ".slip"), "w") as fh:
        s = g["slip_" + ID]
        fh.write("%d\n" % len(s))
        for x in s:
          fh.write("%.17g %.17g %.17g\n" % tuple(x))
  deck = os.path.join(tmp_path, "deck.dat")
  with open(deck, "w") as fh:
    fh.write(str(g["deck"]).replace("output_name                              run",
                                    "output_name                              " + os.path.join(tmp_path, "run")))
  return deck


def replay(g, tmp_path, device, ctx):
  """Returns (integrator, worst location error relative to the largest displacement, worst quaternion error)."""
  from rigidmultiblobswall_amd.read_input import ReadInput
  from rigidmultiblobswall_amd import rigid_integrator, structures
  read = ReadInput(write_case(g, str(tmp_path)))
  integ = rigid_integrator.integrator_from_input(read, device=device, ctx=ctx)
  rigid_integrator.run(read, integ)
  worst_x = worst_q = 0.0
  for ID in [str(x) for x in g["IDs"]]:
    tl, tq = g["trajectory_locations_" + ID], g["trajectory_quaternions_" + ID]
    scale = max(np.abs(tl[-1] - tl[0]).max(), 1e-300)
    for step in range(len(tl)):
      n, loc, quat = structures.read_clones_file(os.path.join(str(tmp_path), "run.%s.%08d.clones" % (ID, step)))
      worst_x = max(worst_x, np.abs(loc - tl[step]).max() / scale)
      worst_q = max(worst_q, np.abs(quat - tq[step]).max())
  return integ, worst_x, worst_q


def reference_counters(g):
  """The reference driver's `.info` file: invalid configurations, GMRES and Lanczos iteration totals of the run."""
  d = dict(line.split("=") for line in str(g["info"]).strip().split("\n"))
  return {k.strip(): int(v) for k, v in d.items()}
