"""`python -m rigidmultiblobswall_amd --input-file deck` -- the two command lines of the reference in one:
`python multi_bodies.py --input-file deck` (time integration, multi_bodies/multi_bodies.py:1107-1118) and
`python multi_bodies_utilities.py --input-file deck` (one-shot schemes).  The scheme named in the deck selects the
engine: `*_rollers` -> rollers.py, `mobility` / `resistance` / `body_mobility` -> utilities.py, anything else ->
rigid_integrator.py.  Backend strings in the deck (`pycuda`, `numba`, ...) are accepted and served by the HIP path."""
import argparse
import shutil
import sys
import time


def main(argv=None):
  ap = argparse.ArgumentParser(prog="python -m rigidmultiblobswall_amd",
                               description="Run a RigidMultiblobsWall input deck on one MI355X, or on several from one process (--devices)")
  ap.add_argument("--input-file", dest="input_file", type=str, default="data.main", help="name of the input file")
  ap.add_argument("--print-residual", action="store_true", help="print gmres and lanczos residuals")
  ap.add_argument("--device", default="cuda:0")
  ap.add_argument("--devices", default=None, metavar="0,1,...",
                  help="run the O(N^2) pair sweeps of the deck on these GPUs from this ONE process (single-process "
                       "multi-device engine, multi.MultiContext); the O(N) solver / stepper logic stays on the first")
  args = ap.parse_args(argv)
  from .read_input import ReadInput
  read = ReadInput(args.input_file)
  shutil.copyfile(args.input_file, read.output_name + ".inputfile")
  t0 = time.time()
  ctx = None
  if args.devices:
    from .multi import MultiContext
    devs = [int(x) for x in args.devices.split(",") if x.strip() != ""]
    ctx = MultiContext(devs)
    args.device = "cuda:%d" % devs[0]
  if read.scheme in ("mobility", "resistance", "body_mobility"):
    from . import utilities
    utilities.run(read, device=args.device, ctx=ctx)
  else:
    if read.scheme.find("rollers") > -1:
      from . import rollers as engine
    else:
      from . import rigid_integrator as engine
    integ = engine.integrator_from_input(read, device=args.device, ctx=ctx, rng=read.random_generator(save=True))
    integ.print_residual = args.print_residual
    with open(read.output_name + ".bodies_info", "w") as fh:
      fh.write("num_of_body_types  %d\n" % len(integ.body_types))
      fh.write("body_names         %s\n" % str(integ.structures_ID))
      fh.write("body_types         %s\n" % str(integ.body_types))
      fh.write("num_bodies         %d\n" % sum(integ.body_types))
      fh.write("num_blobs          %d\n" % integ.Nblobs)

    def progress(step, it):
      if step % read.n_save == 0:
        print("Integrator = ", read.scheme, ", step = ", step, ", invalid configurations", it.invalid_configuration_count,
              ", wallclock time = ", time.time() - t0, flush=True)
    engine.run(read, integ, callback=progress)
    with open(read.output_name + ".info", "w") as fh:
      fh.write("invalid_configuration_count    = %d\n" % integ.invalid_configuration_count)
      fh.write("deterministic_iterations_count = %d\n" % integ.det_iterations_count)
      fh.write("stochastic_iterations_count    = %d\n" % integ.stoch_iterations_count)
      fh.write("nonlinear_iterations_count     = 0\n")
  with open(read.output_name + ".time", "w") as fh:
    fh.write(str(time.time() - t0) + "\n")
  print("\n\n\n# End")
  return 0


if __name__ == "__main__":
  sys.exit(main())
