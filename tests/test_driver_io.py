"""MatrixMarket / parameter-file I/O of the driver (rails_amd/mmio.py, CPU) and the command-line driver end to end on the GPU
(rails_amd/main.py): file names and formats of the reference's driver (src/main.cpp:57-68,111,123-126)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_coordinate_files_general_symmetric_pattern_duplicates(tmp_path):
    from rails_amd import mmio

    p = tmp_path / "g.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n% a comment\n\n3 4 5\n1 1 1.5\n3 4 -2\n1 1 0.5\n2 3 7e-1\n3 1 4\n")
    m, n, rp, col, val = mmio.read_csr(str(p))
    assert (m, n) == (3, 4)
    assert rp.tolist() == [0, 1, 2, 4] and col.tolist() == [0, 2, 0, 3]
    np.testing.assert_allclose(val, [2.0, 0.7, 4.0, -2.0])  # duplicates (1,1) summed, rows sorted by column
    D = mmio.read_dense(str(p))
    assert D.shape == (3, 4) and D[0, 0] == 2.0 and D[2, 3] == -2.0 and D[1, 1] == 0.0
    p.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 3\n1 1 2\n2 1 -1\n3 2 5\n")
    D = mmio.read_dense(str(p))
    np.testing.assert_array_equal(D, [[2, -1, 0], [-1, 0, 5], [0, 5, 0]])
    p.write_text("%%MatrixMarket matrix coordinate real skew-symmetric\n2 2 1\n2 1 3\n")
    np.testing.assert_array_equal(mmio.read_dense(str(p)), [[0, -3], [3, 0]])
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n2 2 2\n1 2\n2 1\n")
    np.testing.assert_array_equal(mmio.read_dense(str(p)), [[0, 1], [1, 0]])
    p.write_text("%%MatrixMarket matrix coordinate real general\n2 2 0\n")
    m, n, rp, col, val = mmio.read_csr(str(p))
    assert rp.tolist() == [0, 0, 0] and val.size == 0
    for bad in ("not a banner\n1 1 1\n", "%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n",
                "%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1\n", "%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1\n"):
        p.write_text(bad)
        with pytest.raises(mmio.MatrixMarketError):
            mmio.read_csr(str(p))


def test_array_and_csr_round_trips_are_bit_exact(tmp_path):
    from rails_amd import mmio
    from rails_amd import problems as P

    g = np.random.default_rng(1)
    V = g.standard_normal((37, 5)) * 10.0 ** g.integers(-12, 12, (37, 5))
    mmio.write_array(str(tmp_path / "V.mtx"), V, comment="two\nlines")
    kind, back = mmio.read(str(tmp_path / "V.mtx"))
    assert kind == "dense" and np.array_equal(back, V)  # %.17g round-trips fp64
    text = (tmp_path / "V.mtx").read_text().splitlines()
    assert text[0] == "%%MatrixMarket matrix array real general" and text[3] == "37 5"
    assert float(text[4]) == V[0, 0] and float(text[5]) == V[1, 0]  # column-major
    A = P.stencil27(5, 4, 3, random_values=True, seed=2)
    m = A[0].size - 1
    mmio.write_csr(str(tmp_path / "A.mtx"), m, m, *A)
    m2, n2, rp, col, val = mmio.read_csr(str(tmp_path / "A.mtx"))
    assert (m2, n2) == (m, m) and np.array_equal(rp, A[0]) and np.array_equal(col, A[1]) and np.array_equal(val, A[2])
    # a dense array file read as an operator
    D = np.array([[1.0, 0.0], [2.5, -3.0]])
    mmio.write_array(str(tmp_path / "D.mtx"), D)
    m2, n2, rp, col, val = mmio.read_csr(str(tmp_path / "D.mtx"))
    assert rp.tolist() == [0, 1, 3] and col.tolist() == [0, 0, 1] and val.tolist() == [1.0, 2.5, -3.0]


def test_parameter_files(tmp_path):
    from rails_amd import mmio

    x = tmp_path / "params.xml"
    x.write_text("""<ParameterList name="RAILS">
  <Parameter name="Something else" type="string" value="ignored"/>
  <ParameterList name="Lyapunov Solver">
    <Parameter name="Restart size" type="int" value="64"/>
    <Parameter name="Tolerance" type="double" value="1e-4"/>
    <Parameter name="Minimize solution space" type="bool" value="false"/>
    <Parameter name="Restart from solution" type="bool" value="true"/>
  </ParameterList>
</ParameterList>""")
    prm = mmio.read_parameters(str(x))
    assert prm == {"Restart size": 64, "Tolerance": 1e-4, "Minimize solution space": 0.0, "Restart from solution": 1.0}
    j = tmp_path / "params.json"
    j.write_text('{"Lyapunov Solver": {"Expand size": 3, "Tolerance": 0.001, "Minimize solution space": true}}')
    assert mmio.read_parameters(str(j)) == {"Expand size": 3, "Tolerance": 0.001, "Minimize solution space": 1.0}
    flat = tmp_path / "flat.xml"
    flat.write_text('<ParameterList><Parameter name="Lanczos iterations" type="int" value="12"/></ParameterList>')
    assert mmio.read_parameters(str(flat)) == {"Lanczos iterations": 12}


@pytest.mark.gpu
def test_driver_end_to_end(tmp_path, oracle):
    from rails_amd import mmio
    from rails_amd import problems as P

    A = P.laplace7(12, 10, 8)
    m = A[0].size - 1
    B = P.rhs(m, 6, seed=4)
    mmio.write_csr(str(tmp_path / "A.mtx"), m, m, *A)
    Bs = B.copy()
    Bs[np.abs(Bs) < 0.05] = 0.0  # B.mtx is a sparse (coordinate) file in the reference's driver
    r, c = np.nonzero(Bs)
    mmio.write_csr(str(tmp_path / "B.mtx"), m, 6, *mmio.coo_to_csr(m, 6, r, c, Bs[r, c]))
    (tmp_path / "params.xml").write_text("""<ParameterList name="p"><ParameterList name="Lyapunov Solver">
<Parameter name="Restart size" type="int" value="80"/><Parameter name="Reduced size" type="int" value="40"/>
<Parameter name="Expand size" type="int" value="6"/><Parameter name="Lanczos iterations" type="int" value="8"/>
<Parameter name="Tolerance" type="double" value="1e-6"/></ParameterList></ParameterList>""")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "rails_amd.main", str(tmp_path / "params.xml"), "--dir", str(tmp_path), "--quiet"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "Loading matrices" in p.stdout and "wrote" in p.stdout
    V = mmio.read_dense(str(tmp_path / "V.mtx"))
    T = mmio.read_dense(str(tmp_path / "T.mtx"))
    assert V.shape[0] == m and T.shape == (V.shape[1], V.shape[1])
    import scipy.sparse as sp

    As = sp.csr_matrix((A[2], A[1], A[0]), shape=(m, m))
    X = V @ T @ V.T
    R = As @ X + (As @ X.T).T + Bs @ Bs.T
    assert np.linalg.norm(R) / np.linalg.norm(Bs @ Bs.T) < 1e-4
    out = oracle.solve(A, Bs, oracle.params({"Restart size": 80, "Reduced size": 40, "Expand size": 6, "Lanczos iterations": 8, "Tolerance": 1e-6,
                                             "rng_mode": 1, "seed": 1}))
    Xo = out["V"] @ out["T"] @ out["V"].T
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-4
    # warm start from the solution: converges at once (src/LyapunovSolver.hpp:116-123)
    p = subprocess.run([sys.executable, "-m", "rails_amd.main", str(tmp_path / "params.xml"), "--dir", str(tmp_path), "--quiet", "--warm-start", "V.mtx",
                        "--V", "V2.mtx", "--T", "T2.mtx"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:]
    V2 = mmio.read_dense(str(tmp_path / "V2.mtx"))
    assert V2.shape[1] <= V.shape[1]
