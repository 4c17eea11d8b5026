"""Structure / configuration files and stock blob models.

File formats are the reference's (plain text, `#` comments):
  .vertex  first non-blank line: number of blobs [optional extra columns]; then one blob per line
           x y z [radius]                       (read_input/read_vertex_file.py:7-33)
  .clones  first line: number of bodies; then per body  x y z  s p1 p2 p3  (location + quaternion,
           normalised on read)                  (read_input/read_clones_file.py:11-52)
"""
import numpy as np


def _data_lines(path):
  with open(path, "r") as fh:
    for line in fh:
      line = line.split("#", 1)[0].strip()
      if line:
        yield line


def read_vertex_file(path):
  """-> (Nblobs, 3) or (Nblobs, 4) float array (4th column = per-blob radius when present)."""
  lines = list(_data_lines(path))
  n = int(lines[0].split()[0])
  rows = [np.array(l.split(), dtype=np.float64) for l in lines[1:]]
  coor = np.array(rows)
  if len(coor) < n:
    raise ValueError("%s: header announces %d blobs, found %d" % (path, n, len(coor)))
  return coor


def read_clones_file(path):
  """-> (number_of_bodies, locations (Nb,3), quaternions (Nb,4) normalised, (s,p1,p2,p3))."""
  lines = list(_data_lines(path))
  n = int(lines[0].split()[0])
  loc, quat = [], []
  for l in lines[1:n + 1]:
    d = [float(x) for x in l.split()[:7]]
    q = np.array(d[3:7])
    loc.append(d[0:3])
    quat.append(q / np.linalg.norm(q))
  return n, np.array(loc).reshape(-1, 3), np.array(quat).reshape(-1, 4)


def read_slip_file(path):
  """Active slip per blob in the body frame (read_input/read_slip_file.py:7-37): first data line = number of blobs,
  then one `sx sy sz` row per blob.  -> (Nblobs, 3)."""
  lines = list(_data_lines(path))
  n = int(lines[0].split()[0])
  slip = np.array([[float(x) for x in l.split()[:3]] for l in lines[1:]], dtype=np.float64).reshape(-1, 3)
  if len(slip) < n:
    raise ValueError("%s: header announces %d blobs, found %d" % (path, n, len(slip)))
  return slip


def icosahedron_shell(geometric_radius):
  """12-blob shell = vertices of a regular icosahedron with one vertex on +z, the model behind the
  reference's Structures/shell_N_12_*.vertex files (e.g. Rg = 0.7921 for hydrodynamic radius 1)."""
  R = float(geometric_radius)
  zc = R / np.sqrt(5.0)
  rc = 2.0 * R / np.sqrt(5.0)
  pts = [[0.0, 0.0, R]]
  for k in range(5):
    ang = 2.0 * np.pi * k / 5.0 + 0.4 * np.pi
    pts.append([rc * np.cos(ang), rc * np.sin(ang), zc])
  for k in range(5):
    ang = 2.0 * np.pi * k / 5.0 + 0.4 * np.pi + np.pi / 5.0
    pts.append([rc * np.cos(ang), rc * np.sin(ang), -zc])
  pts.append([0.0, 0.0, -R])
  return np.array(pts)


def min_blob_separation(reference_configuration):
  r = np.asarray(reference_configuration)[:, :3]
  d = np.linalg.norm(r[:, None, :] - r[None, :, :], axis=-1)
  return d[np.triu_indices(len(r), 1)].min()


def roller_monolayer(n_bodies, radius=1.0155, phi2d=0.4, height=(1.1, 2.0), seed=0):
  """bodies on a perturbed square lattice at area fraction phi2d, heights in
  [height[0] R, height[1] R], random orientations.  -> locations, quaternions, box length."""
  rng = np.random.RandomState(seed)
  side = int(np.ceil(np.sqrt(n_bodies)))
  cell = np.sqrt(np.pi * radius ** 2 / phi2d)
  ij = np.array([(i, j) for i in range(side) for j in range(side)][:n_bodies], dtype=np.float64)
  loc = np.empty((n_bodies, 3))
  loc[:, :2] = (ij + 0.5) * cell + (rng.rand(n_bodies, 2) - 0.5) * (cell - 2.0 * radius) * 0.9
  loc[:, 2] = radius * (height[0] + (height[1] - height[0]) * rng.rand(n_bodies))
  q = rng.randn(n_bodies, 4)
  q /= np.linalg.norm(q, axis=1)[:, None]
  return loc, q, side * cell
