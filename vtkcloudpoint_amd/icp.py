"""Drop-in mirrors of BaseClass/ICP.cs and the slice of BaseClass/Matrix.cs it uses."""
import numpy as np

from . import _native
from .datamodel import xyz_array
from .runtime import default_context


class MException(Exception):
    """BaseClass/Matrix.cs:710-715."""


class Matrix:
    """Row-major double matrix, BaseClass/Matrix.cs:18-34 (storage + indexer), :286-293 (Transpose),
    :500-561 (multiply/add/trace), :692-705 (operators).  Only what ICP's R / T carriers need."""

    def __init__(self, iRows, iCols):
        self.rows, self.cols = int(iRows), int(iCols)
        self.mat = [0.0] * (self.rows * self.cols)

    def __getitem__(self, rc):
        r, c = rc
        return self.mat[r * self.cols + c]  # flat-array bounds only, like the C# (Matrix.cs:30-34)

    def __setitem__(self, rc, v):
        r, c = rc
        self.mat[r * self.cols + c] = float(v)

    @staticmethod
    def ZeroMatrix(r, c):
        return Matrix(r, c)

    @staticmethod
    def IdentityMatrix(r, c):
        m = Matrix(r, c)
        for i in range(min(r, c)):
            m[i, i] = 1.0
        return m

    def Duplicate(self):
        m = Matrix(self.rows, self.cols)
        m.mat = list(self.mat)
        return m

    @staticmethod
    def Transpose(m):
        t = Matrix(m.cols, m.rows)
        for i in range(m.rows):
            for j in range(m.cols):
                t[j, i] = m[i, j]
        return t

    @staticmethod
    def Multiply(a, b):
        if isinstance(a, (int, float)):
            r = Matrix(b.rows, b.cols)
            r.mat = [v * a for v in b.mat]
            return r
        if a.cols != b.rows:
            raise MException("Wrong dimension of matrix!")
        r = Matrix(a.rows, b.cols)
        for i in range(a.rows):
            for j in range(b.cols):
                s = 0.0
                for k in range(a.cols):
                    s += a[i, k] * b[k, j]  # StupidMultiply, Matrix.cs:500-510
                r[i, j] = s
        return r

    @staticmethod
    def Add(a, b):
        if a.rows != b.rows or a.cols != b.cols:
            raise MException("Matrices must have the same dimensions!")
        r = Matrix(a.rows, a.cols)
        r.mat = [x + y for x, y in zip(a.mat, b.mat)]
        return r

    @staticmethod
    def TR(m):
        return sum(m[i, i] for i in range(m.rows))

    def __add__(self, o):
        return Matrix.Add(self, o)

    def __sub__(self, o):
        return Matrix.Add(self, Matrix.Multiply(-1, o))

    def __mul__(self, o):
        return Matrix.Multiply(self, o)

    def __rmul__(self, n):
        return Matrix.Multiply(n, self)

    def to_numpy(self):
        return np.array(self.mat).reshape(self.rows, self.cols)


class ICP:
    """BaseClass/ICP.cs:8-314.  go_hell_ICP keeps the C#'s signature and in-place outputs (R 3x3, T 3x1).
    The arithmetic is the INTENDED Besl-McKay / Horn loop: the as-written C# is non-functional (integer
    division :53, '+' at :66, delta index :76, Jacobi indexing Matrix.cs:636-666, i<9 loop :170-174)."""

    max_iter = 1000  # the C# loops until |d - pre_d| < e with no bound; this is the safety net

    def __init__(self, ctx=None):
        self._ctx = ctx
        self.last = None

    def go_hell_ICP(self, model, data, R, T, e):
        if R.rows != 3 or R.cols != 3 or T.rows != 3 or T.cols != 1:
            raise MException("R must be 3x3 and T 3x1")
        ctx = self._ctx or default_context()
        if len(data) == 0:
            return
        r = ctx.icp(xyz_array(model), xyz_array(data), float(e), self.max_iter, _native.STOP_SSE_DELTA)
        self.last = r
        if r["iters"] == 1 and r["sse"] < e:
            return  # the C# never writes R, T when the very first round already satisfies the stop rule
        for i in range(3):
            for j in range(3):
                R[i, j] = r["R"][i, j]
            T[i, 0] = r["T"][i]

    # the individually-correct sub-functions, kept for source compatibility (host side, tiny inputs)
    @staticmethod
    def CalculateRotation(q, R):
        """ICP.cs:274-285 (Matrix form) / :183-194 (array form)."""
        g = (lambda i: q[i, 0]) if isinstance(q, Matrix) else (lambda i: q[i])
        q0, q1, q2, q3 = g(0), g(1), g(2), g(3)
        vals = [q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, 2.0 * (q1 * q2 - q0 * q3), 2.0 * (q1 * q3 + q0 * q2),
                2.0 * (q1 * q2 + q0 * q3), q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3, 2.0 * (q2 * q3 - q0 * q1),
                2.0 * (q1 * q3 - q0 * q2), 2.0 * (q2 * q3 + q0 * q1), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3]
        for k, v in enumerate(vals):
            if isinstance(R, Matrix):
                R[k // 3, k % 3] = v
            else:
                R[k] = v

    def FindClosestPointSet(self, model, data):
        """ICP.cs:224-250 on the GPU: returns the list Y of matched model points."""
        ctx = self._ctx or default_context()
        if len(model) == 0:
            raise IndexError("model[0] (ICP.cs:233)")
        if len(data) == 0:
            return []
        _, nn = ctx.icp_sums(xyz_array(model), xyz_array(data))
        return [model[int(j)] for j in nn]
