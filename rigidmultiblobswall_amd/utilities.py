"""One-shot problems of multi_bodies/multi_bodies_utilities.py driven by the same decks (BASELINE.json configs[0] is
its `body_mobility` scheme on multi_bodies/inputfile_body_mobility.dat):

  scheme mobility       solve [M -K; -K^T 0][lambda; U] = [slip; -F]  -> <output>.velocity.dat, <output>.force.dat   (:450-547)
  scheme resistance     lambda = M^-1 (slip + K U)                   -> <output>.force.dat                          (:550-581)
  scheme body_mobility  N = (K^T M^-1 K)^+  and  N K^T M^-1          -> <output>.body_mobility.dat, .body_slip_mobility.dat (:583-614)

plus `velocity_field` = the grid evaluation of `plot_velocity_field` (:74-186) through the source->target kernel, with
a plain legacy-VTK ASCII writer standing where the reference calls its compiled visit writer.

Files are written with np.savetxt(delimiter='  ') as the reference does.  `mobility` uses the device-resident GMRES
(rigid.py); the two dense schemes build the dense blob mobility on the device (body_dense_tt_kernel with the whole
suspension as one "body") and factor it with torch on the device -- they are O(N^3) by definition and meant for
N up to a few thousand blobs, like the reference's.
"""
import numpy as np
import torch

from .rigid import RigidSuspension
from .rigid_integrator import bodies_from_input, lab_frame_slip, RigidIntegrator


def _dense_blob_mobility(rs):
  return rs.dense_blob_mobility()


def _dense_K(rs):
  return rs.dense_K()


def run(read, device="cuda:0", ctx=None, write=True):
  """Returns a dict with the arrays of the scheme (numpy) and writes the reference's output files."""
  scheme = read.scheme
  if scheme not in ("mobility", "resistance", "body_mobility"):
    raise ValueError("scheme must be mobility, resistance or body_mobility (got %s)" % scheme)
  b = bodies_from_input(read)
  nb = len(b["refs"])
  wall = read.domain != "no_wall" and "no_wall" not in read.mobility_blobs_implementation
  if scheme == "mobility":
    wall = read.domain != "no_wall" and "no_wall" not in read.mobility_vector_prod_implementation
  out = {}
  if scheme == "mobility":
    integ = RigidIntegrator(b["refs"], b["locations"], b["quaternions"], "deterministic_forward_euler", read.blob_radius,
                            read.eta, tolerance=read.solver_tolerance, domain="single_wall" if wall else "no_wall",
                            periodic_length=read.periodic_length, device=device, ctx=ctx, prescribed=b["prescribed"])
    rs = integ.susp
    if b["slips"] is not None:
      integ.slip_body_frame = torch.as_tensor(b["slips"], device=rs.device)
    slip = integ._slip()
    if read.force_file is not None:
      FT = torch.as_tensor(np.loadtxt(read.resolve(read.force_file)).reshape(nb, 6), device=rs.device)
    else:
      integ.g = read.g
      integ.repulsion_strength_wall, integ.debye_length_wall = read.repulsion_strength_wall, read.debye_length_wall
      if read.blob_blob_force_implementation != "None":
        integ.repulsion_strength, integ.debye_length = read.repulsion_strength, read.debye_length
      FT = integ.force_torque_calculator()
    rhs = rs.prescribe(torch.cat([slip, -FT.reshape(-1)]))
    sol, info = rs.solve(rhs, tol=read.solver_tolerance, restart=60, maxiter=1000)
    sol = rs.impose_prescribed_velocity(sol)
    n3 = 3 * rs.n_blobs
    out["velocity"] = sol[n3:].view(nb, 6).cpu().numpy()
    out["lambda_blobs"] = sol[:n3].view(-1, 3).cpu().numpy()
    out["force"] = rs.KT_times_lambda(sol[:n3].contiguous()).view(nb, 6).cpu().numpy()
    out["r_vectors"] = rs.r_vectors
    out["info"] = info
    if write:
      np.savetxt(read.output_name + ".velocity.dat", out["velocity"], delimiter="  ")
      np.savetxt(read.output_name + ".force.dat", out["force"], delimiter="  ")
    plot = read.options.get("plot_velocity_field")
    if plot:
      grid = np.array(plot.split(), dtype=np.float64)
      tracer = float(read.options.get("tracer_radius") or 0.0)
      out["grid_coor"], out["grid_velocity"] = velocity_field(grid, rs.r_vectors, out["lambda_blobs"], read.blob_radius,
                                                              read.eta, tracer, wall=wall,
                                                              output=read.output_name if write else None)
    if ctx is None:
      integ.close()
    return out

  rs = RigidSuspension(b["refs"], b["locations"], b["quaternions"], read.blob_radius, read.eta, wall=wall,
                       device=device, ctx=ctx)
  M = _dense_blob_mobility(rs)
  K = _dense_K(rs)
  if scheme == "resistance":
    velocity = np.zeros((nb, 6))
    if read.velocity_file is not None:
      velocity = np.loadtxt(read.resolve(read.velocity_file)).reshape(nb, 6)
    slip = torch.zeros(3 * rs.n_blobs, dtype=torch.float64, device=rs.device)
    if b["slips"] is not None:
      slip = lab_frame_slip(rs, torch.as_tensor(b["slips"], device=rs.device))
    rhs = slip + K @ torch.as_tensor(velocity.reshape(-1), device=rs.device)
    lam = torch.linalg.solve(M, rhs)
    out["lambda_blobs"] = lam.view(-1, 3).cpu().numpy()
    out["force"] = (K.t() @ lam).view(nb, 6).cpu().numpy()
    if write:
      np.savetxt(read.output_name + ".force.dat", out["force"], delimiter="  ")
  else:
    R = torch.linalg.inv(M)
    N = torch.linalg.pinv(K.t() @ R @ K)
    out["body_mobility"] = N.cpu().numpy()
    out["body_slip_mobility"] = (N @ (K.t() @ R)).cpu().numpy()
    if write:
      np.savetxt(read.output_name + ".body_mobility.dat", out["body_mobility"], delimiter="  ")
      np.savetxt(read.output_name + ".body_slip_mobility.dat", out["body_slip_mobility"], delimiter="  ")
  out["r_vectors"] = rs.r_vectors
  if ctx is None:
    rs.close()
  return out


def velocity_field(grid, r_vectors_blobs, lambda_blobs, blob_radius, eta, tracer_radius=0.0, wall=True, output=None,
                   radius_blobs=None):
  """Fluid velocity on a rectilinear grid of tracers (multi_bodies_utilities.py:74-186).  grid = 9 numbers
  (x0 x1 nx  y0 y1 ny  z0 z1 nz); x is the fast axis.  Returns (grid_coor (n,3), velocity (n,3)); with `output`
  also writes `<output>.velocity_field.vtk` (legacy VTK, rectilinear grid, cell data `velocity`)."""
  from . import mobility as mob
  grid = np.reshape(np.asarray(grid, dtype=np.float64), (3, 3)).T
  length = grid[1] - grid[0]
  points = np.array(grid[2], dtype=np.int32)
  dx = length / points
  gx = grid[0, 0] + dx[0] * (np.arange(points[0]) + 0.5)
  gy = grid[0, 1] + dx[1] * (np.arange(points[1]) + 0.5)
  gz = grid[0, 2] + dx[2] * (np.arange(points[2]) + 0.5)
  zz, yy, xx = np.meshgrid(gz, gy, gx, indexing="ij")
  coor = np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], axis=1)
  r = np.asarray(r_vectors_blobs, dtype=np.float64).reshape(-1, 3)
  radius_source = np.ones(len(r)) * blob_radius if radius_blobs is None else np.asarray(radius_blobs, dtype=np.float64)
  radius_target = np.ones(len(coor)) * tracer_radius
  fn = (mob.single_wall_mobility_trans_times_force_source_target_hip if wall
        else mob.no_wall_mobility_trans_times_force_source_target_hip)
  vel = fn(r, coor, np.asarray(lambda_blobs, dtype=np.float64).reshape(-1), radius_source, radius_target, eta).reshape(-1, 3)
  if output is not None:
    edges = [np.concatenate([g - 0.5 * d, [hi]]) for g, d, hi in zip((gx, gy, gz), dx, grid[1])]
    with open(output + ".velocity_field.vtk", "w") as fh:
      fh.write("# vtk DataFile Version 2.0\nvelocity field\nASCII\nDATASET RECTILINEAR_GRID\n")
      fh.write("DIMENSIONS %d %d %d\n" % (points[0] + 1, points[1] + 1, points[2] + 1))
      for name, e in zip("XYZ", edges):
        fh.write("%s_COORDINATES %d double\n%s\n" % (name, len(e), " ".join("%.12g" % v for v in e)))
      fh.write("CELL_DATA %d\nVECTORS velocity double\n" % len(coor))
      for v in vel:
        fh.write("%.12g %.12g %.12g\n" % tuple(v))
  return coor, vel
