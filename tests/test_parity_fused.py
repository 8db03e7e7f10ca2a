"""GPU parity of the fused stage kernel (artemis_hip_stage_fused): one launch per RK stage,
primitives ping-ponged between two buffers, conserved state rebuilt in registers.  Bar:
bit-exact against the CPU oracle's full step (which follows the reference's unfused task
order), for every Riemann solver, PCM/PLM, 1-D/2-D/3-D, ragged tiles, several k-chunks."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.oracle import Oracle
from test_parity_ops import random_state, same

pytestmark = pytest.mark.gpu

COEFF = {"rk1": [(0.0, 1.0, 1.0)],
         "rk2": [(0.0, 1.0, 1.0), (0.5, 0.5, 0.5)],
         "vl2": [(0.0, 1.0, 0.5), (0.0, 1.0, 1.0)],
         "rk3": [(0.0, 1.0, 1.0), (0.25, 0.75, 0.25), (2.0 / 3.0, 1.0 / 3.0, 2.0 / 3.0)]}


def fused_step(mb, bufs, integ, dt, bc, cons_out=False, dt_dev=None, cfl=0.0):
    """bufs: [(tensor, table)] * 3; bufs[0] holds the start-of-step primitives (ghosts
    filled) and receives the end-of-step primitives."""
    from artemis_amd import capi
    L = mb.L
    A, B, Cc = bufs
    stages = COEFF[integ]
    cur = A
    for s, (g0, g1, be) in enumerate(stages):
        last = s == len(stages) - 1
        if last:
            out = A  # may alias prim_u1: cell-wise access only
        else:
            out = B if cur is not B else Cc
        mb.stage_fused(g0, g1, be * dt, be * dt, cur[1], A[1], out[1],
                       cons_out=mb.pack.gas.cons0 if (cons_out and last) else None,
                       pcm=(integ == "vl2" and s == 0), cfl=cfl,
                       dt_dev=dt_dev if last else None)
        pk = mb.pack_with_prim(out[1])
        flat = []
        for row in bc:
            flat += [capi.BCS[x] for x in row]
        mb.call_on(pk, L.artemis_hip_apply_bc, (C.c_int * len(flat))(*flat), None)
        cur = out


def setup(nx, ng, recon, riem, bc, seed, gamma=1.4, blast=False, integ="rk2"):
    from artemis_amd.pack import MeshBlockPack
    kw = dict(ng=ng, reconstruct=recon, riemann=riem, gamma=gamma, dfloor=1e-10, siefloor=1e-10)
    o = Oracle(nx, (-1.0, -0.7, 0.1), (1.0, 0.9, 1.3), cfl=0.3, bc=bc, integrator=integ, **kw)
    if blast:
        o.pgen_blast(radius=0.3, internal_energy=1.0, p0=1e-5, d0=1.0, x0=(0.0, 0.1, 0.7), samples=0)
    else:
        random_state(o, np.random.default_rng(seed), mach=1.0, contrast=30.0)
        o.ApplyBoundaryConditions()
        o.PrimToCons()
    mb = MeshBlockPack(1, nx, [(-1.0, -0.7, 0.1)], [(1.0, 0.9, 1.3)], with_fluxes=False, **kw)
    mb.gas_prim[0].copy_(torch.from_numpy(o.gprim.copy()))
    bufs = [(mb.gas_prim, mb.pack.gas.prim), mb.new_prim_buffer("B"), mb.new_prim_buffer("C")]
    return o, mb, bufs


CASES = [
    ((40, 20, 36), 2, "plm", "hllc", "outflow"),
    ((40, 20, 36), 2, "plm", "hlle", "periodic"),
    ((40, 20, 36), 2, "plm", "llf", "reflecting"),
    ((33, 9, 17), 2, "pcm", "hllc", "outflow"),    # ragged in every direction
    ((64, 16, 70), 3, "plm", "hllc", "periodic"),  # several k-chunks, ng = 3
    ((70, 23, 1), 2, "plm", "hllc", "outflow"),    # 2-D
    ((97, 1, 1), 2, "plm", "hlle", "periodic"),    # 1-D
]


@pytest.mark.parametrize("nx,ng,recon,riem,bcname", CASES)
@pytest.mark.parametrize("integ", ["rk2", "vl2", "rk3"])
def test_fused_step_matches_oracle(hiplib, nx, ng, recon, riem, bcname, integ):
    bc = (bcname,) * 6
    o, mb, bufs = setup(nx, ng, recon, riem, bc, seed=11, integ=integ)
    for step in range(3):
        dt = o.new_dt()
        o.dt = dt
        o.step()
        fused_step(mb, bufs, integ, dt, [bc])
        mb.PrimToCons()  # materialise P in the ghosts and the conserved state for comparison
        same(mb.gas_prim[0], o.gprim, f"prim after fused step {step}")
        same(mb.gas_u0[0], o.gu0, f"cons after fused step {step}")


@pytest.mark.parametrize("riem", ["hllc", "hlle", "llf"])
@pytest.mark.parametrize("nx", [(40, 20, 36), (48, 24, 1)])
def test_fused_step_with_vanishing_velocities(hiplib, riem, nx):
    """The tuned Cartesian kernel is exact EVERYWHERE (detect-and-redo).  Ahead of a shock
    velocities decay like 1e-40, 1e-80, 1e-160, 1e-320.  The kernel's hand-scheduled divisions are the bits of IEEE
    divisions only while every numerator is zero or at least 2^-969: the product of two velocity differences of 1e-150
    in a limited slope, or a momentum of 1e-305 divided by the density, is not.  Zones whose stencil holds a velocity
    below 2^-200 are therefore not stored by the stage kernel but listed and recomputed by stage_redo_kernel with IEEE
    arithmetic from the untouched input buffer.  On a state with such velocities next to exact zeros and ordinary
    values, two RK2 steps: every bit of the primitives and of the conserved state equals the oracle's -- in 3-D (the
    x3 march) and in 2-D (the one-plane form) -- and the dt the stage reduces on the device is the oracle's."""
    bc = ("outflow",) * 6
    o, mb, bufs = setup(nx, 2, "plm", riem, bc, seed=13)
    rng = np.random.default_rng(3)
    w = o.gprim
    scale = rng.choice([1.0, 0.0, 1e-300, 1e-306, 1e-250, 1e-160, 1e-150, 1e-100, 1e-40], size=w[1].shape,
                       p=[0.3, 0.1, 0.1, 0.1, 0.08, 0.08, 0.08, 0.08, 0.08])
    for v in (1, 2, 3):
        w[v] *= scale
    o.ApplyBoundaryConditions()
    o.PrimToCons()
    mb.gas_prim[0].copy_(torch.from_numpy(o.gprim.copy()))
    for step in range(2):
        dt = o.new_dt()
        o.dt = dt
        o.step()
        fused_step(mb, bufs, "rk2", dt, [bc])
        mb.PrimToCons()
        for got, ref, what in ((mb.gas_prim[0].cpu().numpy(), o.gprim, "prim"), (mb.gas_u0[0].cpu().numpy(), o.gu0, "cons")):
            bad = got != ref
            assert not bad.any(), (f"{what}, step {step}: {np.count_nonzero(bad)} entries differ, the largest of magnitude "
                                   f"{np.abs(ref[bad]).max():.3e} (gpu {got[bad][0]:.17e} ref {ref[bad][0]:.17e})")


def test_fused_step_with_vanishing_velocities_cons_and_dt(hiplib):
    """The same regime with the last stage also storing the conserved state and reducing the CFL timestep on the device:
    the zones the exact kernel recomputes store `cons` and contribute to dt like the others."""
    bc = ("outflow",) * 6
    o, mb, bufs = setup((40, 20, 36), 2, "plm", "hllc", bc, seed=21)
    rng = np.random.default_rng(5)
    w = o.gprim
    scale = rng.choice([1.0, 0.0, 1e-300, 1e-250, 1e-160, 1e-100], size=w[1].shape, p=[0.5, 0.1, 0.1, 0.1, 0.1, 0.1])
    for v in (1, 2, 3):
        w[v] *= scale
    o.ApplyBoundaryConditions()
    o.PrimToCons()
    mb.gas_prim[0].copy_(torch.from_numpy(o.gprim.copy()))
    dt_dev = torch.empty(1, dtype=torch.float64, device="cuda")
    for step in range(2):
        dt = o.new_dt()
        o.dt = dt
        o.step()
        dt_dev.fill_(torch.finfo(torch.float64).max)
        fused_step(mb, bufs, "rk2", dt, [bc], cons_out=True, dt_dev=C.c_void_p(dt_dev.data_ptr()), cfl=0.3)
        I = np.s_[:, o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1]
        same(mb.gas_prim[0][I], o.gprim[I], f"prim (interior) after step {step}")
        same(mb.gas_u0[0][I], o.gu0[I], f"cons_out (interior) after step {step}")
        assert dt_dev.item() == o.new_dt(), "fused EstimateTimestepMesh"


def test_hint_words_report_vanishing_velocities_to_the_next_stage(hiplib):
    """artemis_stage_args_t.tiny_in / tiny_out / tiny_clear: a ring of one word per stage through which a launch tells
    the next one whether the state it wrote holds a velocity below 2^-200 anywhere; a zero lets the next launch skip the
    per-zone detection.  With vanishing velocities in the state (stage 1 runs with tiny_in = NULL: detect), the launch
    reports 1, stage 2 -- given that word -- detects and the step equals the oracle bit for bit; without any, the words
    stay zero, both stages run without detection, and the step equals the oracle as well.  Each stage's tiny_in word is
    cleared behind it (tiny_clear) so that the same two pointers serve every step."""
    from artemis_amd import capi
    bc = ("outflow",) * 6
    words = torch.zeros(4, dtype=torch.int32, device="cuda")
    W = lambda q: words.data_ptr() + 4 * q
    flat = [capi.BCS[x] for x in bc]
    for tiny in (True, False):
        o, mb, bufs = setup((40, 20, 36), 2, "plm", "hllc", bc, seed=5)
        if tiny:
            rng = np.random.default_rng(8)
            w = o.gprim
            scale = rng.choice([1.0, 0.0, 1e-300, 1e-160, 1e-100], size=w[1].shape, p=[0.6, 0.1, 0.1, 0.1, 0.1])
            for v in (1, 2, 3):
                w[v] *= scale
            o.ApplyBoundaryConditions()
            o.PrimToCons()
            mb.gas_prim[0].copy_(torch.from_numpy(o.gprim.copy()))
        dt = o.new_dt()
        o.dt = dt
        o.step()
        A, B, _ = bufs
        words.zero_()
        # stage 1: the caller knows nothing about its input (tiny_in = NULL: detect); reports into word 1
        mb.stage_fused(0.0, 1.0, dt, dt, A[1], A[1], B[1], tiny_in=None, tiny_out=W(1), tiny_clear=W(0))
        mb.call_on(mb.pack_with_prim(B[1]), mb.L.artemis_hip_apply_bc, (C.c_int * 6)(*flat), None)
        torch.cuda.synchronize()
        assert int(words[1].item()) == (1 if tiny else 0), words
        # stage 2: in = word 1, out = word 0, word 1 cleared behind it
        mb.stage_fused(0.5, 0.5, 0.5 * dt, 0.5 * dt, B[1], A[1], A[1], tiny_in=W(1), tiny_out=W(0), tiny_clear=W(1))
        mb.call_on(mb.pack_with_prim(A[1]), mb.L.artemis_hip_apply_bc, (C.c_int * 6)(*flat), None)
        mb.PrimToCons()
        torch.cuda.synchronize()
        assert int(words[1].item()) == 0 and int(words[0].item()) == (1 if tiny else 0), words
        same(mb.gas_prim[0], o.gprim, "prim after the step (tiny=%s)" % tiny)
        same(mb.gas_u0[0], o.gu0, "cons after the step (tiny=%s)" % tiny)


def test_fused_step_without_redo_shows_the_limit_the_redo_removes(hiplib, monkeypatch, option):
    """ARTEMIS_NO_REDO=1 (the pre-round-3 kernel: every zone stored by the fast path): on the same state a handful of
    values below 1e-120 differ from the oracle -- i.e. the test above is not vacuous, the redo list is what makes it
    exact."""
    option("no_redo", 1)
    bc = ("outflow",) * 6
    o, mb, bufs = setup((40, 20, 36), 2, "plm", "hllc", bc, seed=13)
    rng = np.random.default_rng(3)
    w = o.gprim
    scale = rng.choice([1.0, 0.0, 1e-300, 1e-306, 1e-250, 1e-160, 1e-150, 1e-100, 1e-40], size=w[1].shape,
                       p=[0.3, 0.1, 0.1, 0.1, 0.08, 0.08, 0.08, 0.08, 0.08])
    for v in (1, 2, 3):
        w[v] *= scale
    o.ApplyBoundaryConditions()
    o.PrimToCons()
    mb.gas_prim[0].copy_(torch.from_numpy(o.gprim.copy()))
    dt = o.new_dt()
    o.dt = dt
    o.step()
    fused_step(mb, bufs, "rk2", dt, [bc])
    mb.PrimToCons()  # (materialises the pressure of the ghost zones, as in the test above)
    got, ref = mb.gas_prim[0].cpu().numpy(), o.gprim
    bad = got != ref
    assert bad.any() and np.abs(ref[bad]).max() < 1e-120 and np.count_nonzero(bad) < 1e-3 * ref.size


def test_fused_blast_with_fused_dt_and_cons(hiplib):
    """Sedov deck: the last stage also writes u0 and min-combines the CFL timestep."""
    import math
    bc = ("outflow",) * 6
    o, mb, bufs = setup((48, 40, 32), 2, "plm", "hllc", bc, seed=0, blast=True)
    dt_dev = torch.empty(1, dtype=torch.float64, device="cuda")
    for step in range(5):
        dt = o.new_dt()
        o.dt = dt
        o.step()
        dt_dev.fill_(torch.finfo(torch.float64).max)
        fused_step(mb, bufs, "rk2", dt, [bc], cons_out=True, dt_dev=C.c_void_p(dt_dev.data_ptr()), cfl=0.3)
        I = np.s_[:, o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1]
        same(mb.gas_prim[0][I], o.gprim[I], f"prim (interior) after step {step}")
        same(mb.gas_u0[0][I], o.gu0[I], f"cons_out (interior) after step {step}")
        assert dt_dev.item() == o.new_dt(), "fused EstimateTimestepMesh"
        assert math.isfinite(dt_dev.item())


def test_fused_rejects_unsupported(hiplib):
    from artemis_amd import capi
    o, mb, bufs = setup((16, 16, 16), 3, "ppm", "hllc", ("periodic",) * 6, seed=3)
    with pytest.raises(capi.ArtemisHipError) as e:
        mb.stage_fused(0.0, 1.0, 1e-3, 1e-3, bufs[0][1], bufs[0][1], bufs[1][1])
    assert e.value.code == capi.EUNSUPPORTED
    with pytest.raises(capi.ArtemisHipError) as e:
        mb.stage_fused(0.0, 1.0, 1e-3, 1e-3, bufs[0][1], bufs[0][1], bufs[0][1])
    assert e.value.code == capi.EINVAL


def test_fast_div_sqrt(hiplib):
    """The fused kernel's hand-scheduled division (refined reciprocal shared between divisions)
    and square root return the same bits as the compiler's IEEE-correct a/b and sqrt() for
    operands anywhere within 2^+-400 -- 4e7 random pairs, plus mantissa edge patterns."""
    n = 1 << 22
    g = torch.Generator(device="cuda").manual_seed(1234)
    for rep in range(10):
        mant = torch.rand(2, n, dtype=torch.float64, device="cuda", generator=g) + 1.0
        expo = torch.randint(-400, 400, (2, n), device="cuda", generator=g).to(torch.float64)
        sign = torch.where(torch.rand(2, n, device="cuda", generator=g) < 0.5, -1.0, 1.0).to(torch.float64)
        ab = sign * torch.ldexp(mant, expo.to(torch.int32))
        if rep == 0:  # mantissas of all ones / one ulp above a power of two / exact quotients
            ab[0, :1000] = torch.nextafter(torch.full((1000,), 2.0, dtype=torch.float64, device="cuda"),
                                           torch.zeros(1000, dtype=torch.float64, device="cuda"))
            ab[1, 1000:2000] = torch.nextafter(torch.ones(1000, dtype=torch.float64, device="cuda"),
                                               torch.full((1000,), 2.0, dtype=torch.float64, device="cuda"))
            ab[0, 2000:3000] = 0.0
            ab[0, 3000:4000] = ab[1, 3000:4000] * 3.0
        outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
        rc = hiplib.artemis_hip_selftest_divsqrt(n, *(C.c_void_p(t.data_ptr()) for t in (ab[0], ab[1], *outs)),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        qf, qi, sf, si = outs
        assert torch.equal(qf.view(torch.int64), qi.view(torch.int64)), "fast division != IEEE division"
        assert torch.equal(sf.view(torch.int64), si.view(torch.int64)), "fast sqrt != IEEE sqrt"
        # and the device results equal the host's correctly rounded ones
        if rep == 0:
            a, b = ab[0].cpu().numpy(), ab[1].cpu().numpy()
            assert np.array_equal(qi.cpu().numpy(), a / b)
            assert np.array_equal(si.cpu().numpy(), np.sqrt(np.abs(b)))


@pytest.mark.parametrize("nx", [(96, 32, 24), (100, 30, 9), (40, 20, 36), (70, 23, 1)])
def test_fused_shell_then_bulk_equals_whole(hiplib, nx):
    """region 1 (boundary shell) followed by region 2 (bulk) writes exactly what region 0 writes,
    and the shell alone already contains every cell a neighbour's ghost slab is cut from."""
    bc = ("outflow",) * 6
    o, mb, bufs = setup(nx, 2, "plm", "hllc", bc, seed=21)
    A, B, Cc = bufs
    dt = o.new_dt()
    args = (0.5, 0.5, 0.5 * dt, 0.5 * dt, A[1], A[1])
    mb.stage_fused(*args, B[1], region=0)
    for faces in (0b110000, 0b001100, 0b100101):  # any subset of faces: 1 then 2 still == 0
        Cc[0].zero_()
        mb.stage_fused(*args, Cc[1], region=1, shell_faces=faces)
        mb.stage_fused(*args, Cc[1], region=2, shell_faces=faces)
        torch.cuda.synchronize()
        assert torch.equal(B[0], Cc[0]), bin(faces)
    Cc[0].zero_()
    mb.stage_fused(*args, Cc[1], region=1)
    shell_only = Cc[0].clone()
    mb.stage_fused(*args, Cc[1], region=2)
    torch.cuda.synchronize()
    assert torch.equal(B[0], Cc[0])
    g = 2
    I = (slice(None), slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    whole, shell = B[0][I], shell_only[I]
    for d in range(o.ndim):  # the nghost layers next to every face are final after region 1
        ax = 4 - d
        n = whole.shape[ax]
        lo = [slice(None)] * 5
        hi = [slice(None)] * 5
        lo[ax], hi[ax] = slice(0, g), slice(n - g, n)
        assert torch.equal(whole[tuple(lo)], shell[tuple(lo)])
        assert torch.equal(whole[tuple(hi)], shell[tuple(hi)])


@pytest.mark.parametrize("general", [False, True])
def test_fused_stage_with_active_floors(hiplib, general):
    """A step far beyond the CFL limit drives densities and energies through the floors inside the
    stage (negative updated density, E < KE, tiny internal energy): the fused epilogues
    (SetAuxillaryFields + ConsToPrim in registers) must floor exactly like the task chain -- tuned
    kernel and general cell-centred stage."""
    from artemis_amd.pack import MeshBlockPack
    nx = (40, 20, 12)
    kw = dict(ng=2, reconstruct="plm", riemann="hllc", gamma=1.4, dfloor=1e-2, siefloor=1e-3)
    o = Oracle(nx, (-1.0, -0.7, 0.1), (1.0, 0.9, 1.3), cfl=0.3, bc=("outflow",) * 6, integrator="rk2", **kw)
    random_state(o, np.random.default_rng(71), mach=3.0, contrast=1.0e3)
    o.ApplyBoundaryConditions()
    o.PrimToCons()
    mb = MeshBlockPack(1, nx, [(-1.0, -0.7, 0.1)], [(1.0, 0.9, 1.3)], with_fluxes=False, **kw)
    mb.gas_prim[0].copy_(torch.from_numpy(o.gprim.copy()))
    out, tout = mb.new_prim_buffer("o")
    dt = 40.0 * o.new_dt()
    o.DeepCopyConservedData()
    o.CalculateFluxes(0, False)
    o.ApplyUpdate(0.0, 1.0, dt)
    o.FluxSource(dt)
    hit = o.gu0[0][o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1]
    assert (hit < 1e-2).mean() > 0.02  # the density floor is really in play
    o.SetAuxillaryFields()
    o.ConsToPrim()
    if general:
        mb.stage_general(0.0, 1.0, dt, dt, gas=(mb.gas_prim_table, mb.gas_prim_table, tout))
    else:
        mb.stage_fused(0.0, 1.0, dt, dt, mb.gas_prim_table, mb.gas_prim_table, tout)
    I = np.s_[:, o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1]
    got, ref = out[0].cpu().numpy()[I], o.gprim[I]
    keep = [0, 1, 2, 3, 5]  # the general stage does not write the pressure slot
    assert np.array_equal(got[keep], ref[keep])


@pytest.mark.parametrize("nx,ng,recon", [((40, 20, 36), 2, "plm"), ((33, 9, 17), 2, "pcm"), ((64, 16, 70), 3, "plm"),
                                         ((70, 23, 1), 2, "plm"), ((97, 1, 1), 2, "plm")])
@pytest.mark.parametrize("tiny", [False, True])
def test_outflow_faces_are_not_read(hiplib, nx, ng, recon, tiny):
    """artemis_stage_args_t.outflow_faces: with a bit set the kernel (and its exact pass: `tiny` puts vanishing velocities
    into the state) stages the edge zone instead of reading the ghost zones behind that face.  Two RK2 steps in which the
    ghost zones of the stage inputs hold NaN and NO boundary fill runs between the stages equal the oracle's steps with
    its outflow conditions after every stage, bit for bit on the active zones -- full tiles, ragged tiles, several
    k-chunks, three ghost zones, 2-D and 1-D; with one face left out of the mask the result changes, i.e. the mask is
    what keeps the ghost zones unread."""
    bc = ("outflow",) * 6
    o, mb, bufs = setup(nx, ng, recon, "hllc", bc, seed=31)
    ndim = 3 if nx[2] > 1 else (2 if nx[1] > 1 else 1)
    if tiny:
        rng = np.random.default_rng(8)
        scale = rng.choice([1.0, 0.0, 1e-300, 1e-250, 1e-160], size=o.gprim[1].shape, p=[0.6, 0.1, 0.1, 0.1, 0.1])
        for v in (1, 2, 3):
            o.gprim[v] *= scale
        o.ApplyBoundaryConditions()
        o.PrimToCons()
        mb.gas_prim[0].copy_(torch.from_numpy(o.gprim.copy()))
    I = np.s_[:, o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1]
    inner = torch.zeros(mb.gas_prim[0].shape[1:], dtype=torch.bool, device="cuda")
    inner[o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1] = True

    def poison(t):  # NaN in every ghost zone of a prim buffer
        t[0][:, ~inner] = float("nan")

    mask = (1 << (2 * ndim)) - 1
    A, B, Cc = bufs
    for step in range(2):
        dt = o.new_dt()
        o.dt = dt
        o.step()
        cur = A
        for s_, (g0, g1, be) in enumerate(COEFF["rk2"]):
            out = A if s_ == 1 else B
            poison(cur[0])
            if s_ == 1:  # (the last stage writes over the start-of-step buffer: its zones are read cell-wise only)
                assert out is A
            mb.stage_fused(g0, g1, be * dt, be * dt, cur[1], A[1], out[1], outflow_faces=mask)
            cur = out
        same(mb.gas_prim[0][I][[0, 1, 2, 3, 5]], o.gprim[I][[0, 1, 2, 3, 5]], f"step {step}")
    # one face missing from the mask: its NaN ghosts are read (the floors of ConsToPrim turn the NaN into floor values)
    poison(A[0])
    mb.stage_fused(0.0, 1.0, 1e-4, 1e-4, A[1], A[1], B[1], outflow_faces=mask)
    mb.stage_fused(0.0, 1.0, 1e-4, 1e-4, A[1], A[1], Cc[1], outflow_faces=mask & ~2)
    assert not torch.equal(B[0][0][I][0], Cc[0][0][I][0])


def test_outflow_faces_block_by_block(hiplib):
    """artemis_stage_args_t.outflow_faces_by_block: one mask per block of the pack.  Two blocks with different states;
    block 0's six faces are all in its mask, block 1's mask leaves its upper x3 face out (a neighbour would fill those
    ghost zones).  With NaN in every ghost zone of block 0 and in every masked ghost zone of block 1 -- its upper x3
    ghost planes hold the outflow values -- both blocks equal their oracle's stage with outflow conditions; with NaN in
    block 1's upper x3 ghost planes as well only block 1 changes: the masks act block by block."""
    from artemis_amd.pack import MeshBlockPack
    nx, ng = (40, 20, 36), 2
    kw = dict(ng=ng, reconstruct="plm", riemann="hllc", gamma=1.4, dfloor=1e-10, siefloor=1e-10)
    los, his = [(-1.0, -0.7, 0.1), (0.2, -0.3, 1.3)], [(1.0, 0.9, 1.3), (2.2, 1.3, 2.5)]
    os_ = []
    for q in range(2):
        o = Oracle(nx, los[q], his[q], cfl=0.3, bc=("outflow",) * 6, integrator="rk2", **kw)
        random_state(o, np.random.default_rng(40 + q), mach=1.0, contrast=30.0)
        o.ApplyBoundaryConditions()
        o.PrimToCons()
        os_.append(o)
    mb = MeshBlockPack(2, nx, los, his, with_fluxes=False, **kw)
    for q in range(2):
        mb.gas_prim[q].copy_(torch.from_numpy(os_[q].gprim.copy()))
    o = os_[0]
    inner = torch.zeros(mb.gas_prim[0].shape[1:], dtype=torch.bool, device="cuda")
    inner[o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1] = True
    upper3 = torch.zeros_like(inner)
    upper3[o.ke + 1:, :, :] = True  # (the upper x3 ghost planes over the entire x1 / x2 extent)
    start = mb.gas_prim.clone()
    B, C2 = mb.new_prim_buffer("B"), mb.new_prim_buffer("C")
    dt = 1.0e-4
    for oo in os_:
        oo.DeepCopyConservedData()
    I = np.s_[:, o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1]
    want = []
    for oo in os_:
        oo.CalculateFluxes(0, False)
        oo.ApplyUpdate(0.0, 1.0, dt)
        oo.FluxSource(dt, 0)
        oo.SetAuxillaryFields()
        oo.ConsToPrim()
        want.append(oo.gprim[I][[0, 1, 2, 3, 5]])
    keep = [0, 1, 2, 3, 5]
    # run 1: block 1's upper x3 ghost planes keep their (outflow) values
    mb.gas_prim[0][:, ~inner] = float("nan")
    mb.gas_prim[1][:, ~inner & ~upper3] = float("nan")
    mb.stage_fused(0.0, 1.0, dt, dt, mb.pack.gas.prim, mb.pack.gas.prim, B[1], outflow_faces_by_block=[63, 31])
    for q in range(2):
        same(B[0][q][I][keep], want[q], f"block {q}")
    # run 2: NaN there too -- block 1 reads them (bit 5 is not in its mask), block 0 is untouched by that
    mb.gas_prim.copy_(start)
    mb.gas_prim[0][:, ~inner] = float("nan")
    mb.gas_prim[1][:, ~inner] = float("nan")
    mb.stage_fused(0.0, 1.0, dt, dt, mb.pack.gas.prim, mb.pack.gas.prim, C2[1], outflow_faces_by_block=[63, 31])
    same(C2[0][0][I][keep], want[0], "block 0 with every mask bit")
    assert not torch.equal(C2[0][1][I][0], B[0][1][I][0])
