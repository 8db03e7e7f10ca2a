"""Multilevel (SMR) CPU oracle: a forest of per-block oracles advanced together.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  Restates, independently of the product's C++ driver,
what Parthenon does around Artemis' tasks on a statically refined mesh.  Parthenon itself is an empty
submodule of /root/reference (version unpinned), so everything marked "upstream, recalled" follows the
published Athena++/Parthenon algorithm; the Artemis-side anchors are cited:

  * block tree + static regions      <parthenon/mesh> refinement = static, <parthenon/static_refinementN>
                                     (inputs/disk/disk_cart.in:42,68-75); region -> logical range and 2:1
                                     balance across faces, edges and corners: upstream, recalled
  * custom refinement operators      RestrictAverage / ProlongateSharedMinMod registered for every field
                                     (utils/artemis_utils.cpp:92-112; restriction.hpp:42-114,
                                     prolongation.hpp:83-184) -- the per-block oracle's C++ restatement
  * what is communicated             FillGhost primitives rho, v, sie (gas.cpp:244-270)
  * stage order                      artemis_driver.cpp:165-266: fluxes -> flux correction (:196-202) ->
                                     ApplyUpdate ... ConsToPrim -> boundary exchange incl. prolongation
                                     (:258, AddBoundaryExchangeTasks with pmesh->multilevel) -> PrimToCons
  * flux correction                  every WithFluxes variable (cons D, M, E, e_int and the pressure flux,
                                     gas.cpp:212-252) and the Metadata::Flux diffusion fluxes (gas.cpp:276-
                                     284) restricted with RestrictAverage<el = F1|F2|F3> (area weights,
                                     restriction.hpp:57-94); gas::face::velocity is neither and is not corrected
  * ghost exchange (upstream, recalled: SendBoundBufs / SetBounds / ApplyBoundaryConditionsOnCoarseOrFine(coarse)
    / ProlongateBounds / ...(fine)): same level copies interior -> ghost over 3^ndim - 1 directions; a finer
    neighbour sends its data restricted; a coarser neighbour's interior lands in the fine block's COARSE BUFFER
    ((nghost+1)/2 + 1 coarse zones deep); the fine block restricts its own interior and the ghost halos it got
    from same-level / finer neighbours into that buffer, applies the physical conditions to it, prolongates into
    the ghost zones facing coarser neighbours, and finally applies the physical conditions on the fine array.

Parity unpinned against a Parthenon build (none exists here); pinned on properties instead: conservation to
round-off across level boundaries, exactness on uniform states, and agreement with the single-level oracle
where the two must coincide (tests/test_multilevel_*.py).
"""
import itertools

import numpy as np

from .oracle import GAS, INTEG, Oracle


class BlockTree:
    """Logical block tree over an nrb[0] x nrb[1] x nrb[2] root grid (upstream, recalled: MeshBlockTree)."""

    def __init__(self, nrb, ndim, periodic):
        self.nrb, self.ndim, self.periodic = tuple(nrb), ndim, tuple(periodic)
        self.internal = set()  # refined nodes (level, (l1, l2, l3))

    def extent(self, level):
        return tuple(self.nrb[d] << level if d < self.ndim else 1 for d in range(3))

    def wrap(self, level, loc):
        """Logical location with periodic images folded back; None outside a non-periodic boundary."""
        ext = self.extent(level)
        out = []
        for d in range(3):
            x = loc[d]
            if x < 0 or x >= ext[d]:
                if not self.periodic[d]:
                    return None
                x %= ext[d]
            out.append(x)
        return tuple(out)

    def parent(self, loc):
        return tuple(loc[d] >> 1 if d < self.ndim else 0 for d in range(3))

    def exists(self, level, loc):
        return level == 0 or (level - 1, self.parent(loc)) in self.internal

    def directions(self):
        r = [(-1, 0, 1) if d < self.ndim else (0,) for d in range(3)]
        return [o for o in itertools.product(*r) if any(o)]

    def ensure(self, level, loc):
        if level > 0 and not self.exists(level, loc):
            self.refine(level - 1, self.parent(loc))

    def refine(self, level, loc):
        if (level, loc) in self.internal:
            return
        self.ensure(level, loc)
        self.internal.add((level, loc))
        for o in self.directions():  # 2:1 balance over faces, edges and corners
            n = self.wrap(level, tuple(loc[d] + o[d] for d in range(3)))
            if n is not None:
                self.ensure(level, n)

    def add_region(self, level, lo, hi, xmin, xmax):
        """<parthenon/static_refinementN>: every block of `level` overlapping [lo, hi] exists (upstream, recalled:
        first block whose right edge exceeds lo ... first whose right edge reaches hi, rounded out to sibling pairs)."""
        rng = []
        for d in range(3):
            if d >= self.ndim:
                rng.append(range(0, 1))
                continue
            n = self.nrb[d] << level
            edge = lambda l: xmin[d] * (1.0 - l / n) + xmax[d] * (l / n) if l < n else xmax[d]
            lmin = next((l for l in range(n) if edge(l + 1) > lo[d]), n - 1)
            lmax = next((l for l in range(lmin, n) if edge(l + 1) >= hi[d]), n - 1)
            if lmin % 2 == 1:
                lmin -= 1
            if lmax % 2 == 0:
                lmax += 1
            rng.append(range(lmin, lmax + 1))
        for l3 in rng[2]:
            for l2 in rng[1]:
                for l1 in rng[0]:
                    self.ensure(level, (l1, l2, l3))

    def leaves(self):
        """Leaves in Z-order (children: x1 fastest)."""
        out = []

        def walk(level, loc):
            if (level, loc) in self.internal:
                for c3 in range(2 if self.ndim > 2 else 1):
                    for c2 in range(2 if self.ndim > 1 else 1):
                        for c1 in range(2):
                            walk(level + 1, (2 * loc[0] + c1, 2 * loc[1] + c2, 2 * loc[2] + c3))
            else:
                out.append((level, loc))

        for l3 in range(self.nrb[2]):
            for l2 in range(self.nrb[1]):
                for l1 in range(self.nrb[0]):
                    walk(0, (l1, l2, l3))
        return out

    def neighbour(self, level, loc, o):
        """('phys', None) | ('same', loc) | ('coarser', parent loc) | ('finer', [(child loc, c), ...])"""
        n = self.wrap(level, tuple(loc[d] + o[d] for d in range(3)))
        if n is None:
            return "phys", None
        if not self.exists(level, n):
            return "coarser", self.parent(n)
        if (level, n) not in self.internal:
            return "same", n
        kids = []
        cr = [((1,) if o[d] < 0 else (0,) if o[d] > 0 else (0, 1)) if d < self.ndim else (0,) for d in range(3)]
        for c in itertools.product(*cr):
            kids.append((tuple(2 * n[d] + c[d] if d < self.ndim else 0 for d in range(3)), c))
        return "finer", kids


class MultiLevelOracle:
    def __init__(self, mesh_nx, block_nx, xmin, xmax, bc, regions=(), ng=2, integrator="rk2", **kw):
        """mesh_nx: root-level zones; block_nx: zones per mesh block; bc: six names; regions: iterable of
        (level, (x1min, x1max), (x2min, x2max), (x3min, x3max)); kw: Oracle keywords (gas package etc.)."""
        self.nx = tuple(block_nx)
        self.ndim = sum(n > 1 for n in mesh_nx)
        self.ng, self.xmin, self.xmax, self.bc = ng, tuple(xmin), tuple(xmax), tuple(bc)
        self.integrator = integrator
        assert ng % 2 == 0, "multilevel meshes need an even number of ghost zones (upstream)"
        nrb = tuple(mesh_nx[d] // block_nx[d] for d in range(3))
        periodic = tuple(bc[2 * d] == "periodic" for d in range(3))
        self.tree = BlockTree(nrb, self.ndim, periodic)
        self.regions = list(regions)
        for level, r1, r2, r3 in regions:
            self.tree.add_region(level, (r1[0], r2[0], r3[0]), (r1[1], r2[1], r3[1]), xmin, xmax)
        self.kw = dict(kw, ng=ng, integrator=integrator)
        self.time, self.dt, self.ncycle = 0.0, None, 0
        self._build_blocks()

    def make_block(self, level, loc):
        """(fine block oracle, its coarse-buffer oracle) of the leaf at a logical location."""
        xmin, xmax, bc = self.xmin, self.xmax, self.bc
        periodic = self.tree.periodic
        lo, hi, bcs = [], [], []
        ext = self.tree.extent(level)
        for d in range(3):
            n = ext[d]
            edge = lambda l: (xmin[d] if l == 0 else xmax[d] if l == n else
                              xmin[d] * (1.0 - l / n) + xmax[d] * (l / n))
            lo.append(edge(loc[d])), hi.append(edge(loc[d] + 1))
            for side in (0, 1):
                outside = (loc[d] == 0) if side == 0 else (loc[d] + 1 == n)
                if d >= self.ndim:
                    bcs.append("outflow")
                elif outside and not periodic[d]:
                    bcs.append(bc[2 * d + side])
                else:
                    bcs.append("none")
        o = Oracle(self.nx, lo, hi, bc=bcs, mesh_bounds=sum(([xmin[d], xmax[d]] for d in range(3)), []), **self.kw)
        cnx = tuple(n // 2 if n > 1 else 1 for n in self.nx)
        return o, Oracle(cnx, lo, hi, bc=bcs, **self.kw)

    def _build_blocks(self, keep=None):
        """One oracle (+ coarse buffer) per leaf of self.tree; `keep` maps leaves to existing (block, coarse) pairs
        that are reused (an adaptive mesh keeps the blocks that do not change)."""
        self.leaves = self.tree.leaves()
        self.index = {lf: b for b, lf in enumerate(self.leaves)}
        self.blocks, self.coarse = [], []
        for lf in self.leaves:
            o, c = keep[lf] if keep and lf in keep else self.make_block(*lf)
            self.blocks.append(o), self.coarse.append(c)
        ns = self.blocks[0].cfg.ns_gas
        self.fill = [v for v in range(6 * ns) if not (4 * ns <= v < 5 * ns)]  # pressure is not FillGhost
        self.has_dust = self.blocks[0].cfg.ns_dust > 0
        self._classify()

    # ---- geometry of the exchange ------------------------------------------------------------------
    def _classify(self):
        t = self.tree
        self.nbr = []
        for level, loc in self.leaves:
            d_ = {}
            for o in t.directions():
                kind, what = t.neighbour(level, loc, o)
                if kind == "same":
                    d_[o] = ("same", self.index[(level, what)], what)
                elif kind == "coarser":
                    d_[o] = ("coarser", self.index[(level - 1, what)], what)
                elif kind == "finer":
                    d_[o] = ("finer", [(self.index[(level + 1, cl)], cl, c) for cl, c in what], None)
                else:
                    d_[o] = ("phys", None, None)
            self.nbr.append(d_)

    def _start(self, d):  # first interior index of dimension d (same for fine arrays and coarse buffers)
        return self.ng if d < self.ndim else 0

    def _ghost_box(self, o, width, n):
        """Index ranges (x1, x2, x3) of the box in direction o: `width` zones outside, n[d] inside extent."""
        out = []
        for d in range(3):
            s = self._start(d)
            if o[d] < 0:
                out.append((s - width, s))
            elif o[d] > 0:
                out.append((s + n[d], s + n[d] + width))
            else:
                out.append((s, s + n[d]))
        return out

    @staticmethod
    def _sl(box, shift=(0, 0, 0)):
        return (slice(None),) + tuple(slice(box[d][0] + shift[d], box[d][1] + shift[d]) for d in (2, 1, 0))

    def _assign(self, dst, box, src, shift=(0, 0, 0)):
        """FillGhost variables of oracle `src` -> oracle `dst`: gas rho, v, sie (gas.cpp:244-270) and every dust
        primitive (dust.cpp:201-213)."""
        a, b = self._sl(box)[1:], self._sl(box, shift)[1:]
        dg, sg = dst.gprim, src.gprim
        for v in self.fill:
            dg[v][a] = sg[v][b]
        if self.has_dust:
            dd, sd = dst.dprim, src.dprim
            for v in range(dd.shape[0]):
                dd[v][a] = sd[v][b]

    def _restrict(self, fine, coarse, crange, S):
        fine.RestrictAverage(coarse, crange, S, S)
        if self.has_dust:
            fine.RestrictAverage(coarse, crange, S, S, field="dust.prim")

    def _prolongate(self, fine, coarse, crange, S):
        fine.ProlongateSharedMinMod(coarse, crange, S, S)
        if self.has_dust:
            fine.ProlongateSharedMinMod(coarse, crange, S, S, field="dust.prim")

    def fill_ghosts(self):
        """Boundary exchange of the FillGhost primitives + physical conditions (module docstring)."""
        nx, ng, t = self.nx, self.ng, self.tree
        cnx = tuple(n // 2 if n > 1 else 1 for n in nx)
        cng = (ng + 1) // 2 + 1
        S = [self._start(d) for d in range(3)]
        span = lambda lo, n: tuple(v for d in range(3) for v in (lo[d], lo[d] + n[d] - 1))
        # 1. every block restricts its interior into its own coarse buffer
        for o, c in zip(self.blocks, self.coarse):
            self._restrict(o, c, span(S, cnx), S)
        # 2. transfers; every source is an interior (fine array or coarse buffer) prepared above
        for b, (level, loc) in enumerate(self.leaves):
            me, mec = self.blocks[b], self.coarse[b]
            for o, (kind, who, what) in self.nbr[b].items():
                if kind == "same":
                    self._assign(me, self._ghost_box(o, ng, nx), self.blocks[who],
                                 tuple(-o[d] * nx[d] for d in range(3)))
                elif kind == "finer":
                    for cb, cl, c in who:
                        box = self._ghost_box(o, ng, nx)
                        shift = []
                        for d in range(3):
                            if d < self.ndim and o[d] == 0:  # the half of my extent this child covers
                                h0 = S[d] + c[d] * cnx[d]
                                box[d] = (h0, h0 + cnx[d])
                            # child's coarse-buffer index of my zone i: -o nx + (i - s) - c nx/2 + s
                            shift.append(-o[d] * nx[d] - c[d] * cnx[d] if d < self.ndim else 0)
                        self._assign(me, box, self.coarse[cb], shift)
                elif kind == "coarser":
                    n_wrapped = t.wrap(level, tuple(loc[d] + o[d] for d in range(3)))
                    shift = []
                    for d in range(3):
                        rel = n_wrapped[d] - o[d]  # my location relative to the (wrapped) neighbour
                        shift.append(rel * cnx[d] - what[d] * nx[d] if d < self.ndim else 0)
                    self._assign(mec, self._ghost_box(o, cng, cnx), self.blocks[who], shift)
        # 3.-5. blocks with a coarser neighbour: restrict the ghost halos that hold fine data, physical
        # conditions on the coarse buffer, prolongate into the zones facing coarser neighbours
        for b in range(len(self.leaves)):
            kinds = self.nbr[b]
            if not any(k[0] == "coarser" for k in kinds.values()):
                continue
            o_, c_ = self.blocks[b], self.coarse[b]
            for o, (kind, _, _) in kinds.items():
                if kind in ("same", "finer"):
                    box = self._ghost_box(o, ng // 2, cnx)
                    self._restrict(o_, c_, tuple(v for d in range(3) for v in (box[d][0], box[d][1] - 1)), S)
            c_.ApplyBoundaryConditions()
            for o, (kind, _, _) in kinds.items():
                if kind == "coarser":
                    box = self._ghost_box(o, ng // 2, cnx)
                    self._prolongate(o_, c_, tuple(v for d in range(3) for v in (box[d][0], box[d][1] - 1)), S)
        # 6. physical conditions on the fine arrays
        for o in self.blocks:
            o.ApplyBoundaryConditions()

    # ---- flux correction (artemis_driver.cpp:196-202) ----------------------------------------------------
    def _flux_sets(self, o, d):
        sets = [o.gflux(d), o.gpflux(d)]
        if self.diffusion:
            sets.append(o.qflux(d))
        if o.cfg.ns_dust:
            sets.append(o.dflux(d))
        return sets

    def flux_correction(self):
        nx, S = self.nx, [self._start(d) for d in range(3)]
        cnx = tuple(n // 2 if n > 1 else 1 for n in nx)
        for b in range(len(self.leaves)):
            for o, (kind, who, _) in self.nbr[b].items():
                if kind != "finer" or sum(abs(x) for x in o) != 1:
                    continue
                d = [abs(x) for x in o].index(1)
                for cb, cl, c in who:
                    f = self.blocks[cb]
                    area = f.face_areas(d + 1)
                    # coarse face slab: index s (my lower face) or s + nx (upper); fine face on the child: the other side
                    dst, srcs = [], []
                    for q in range(3):
                        if q == d:
                            ci = S[q] if o[q] < 0 else S[q] + nx[q]
                            fi = S[q] + nx[q] if o[q] < 0 else S[q]
                            dst.append(slice(ci, ci + 1)), srcs.append([slice(fi, fi + 1)])
                        elif q < self.ndim:
                            h0 = S[q] + c[q] * cnx[q]
                            dst.append(slice(h0, h0 + cnx[q]))
                            srcs.append([slice(S[q] + off, S[q] + nx[q], 2) for off in (0, 1)])
                        else:
                            dst.append(slice(0, 1)), srcs.append([slice(0, 1)])
                    # RestrictAverage<el = F_d>: terms[ok][oj][oi] over the included (tangential) offsets, summed
                    # pairwise as ((t000 + t010) + (t001 + t011)) + ((t100 + t110) + (t101 + t111))
                    def gather(arr):
                        tm = {}
                        for ok in range(2):
                            for oj in range(2):
                                for oi in range(2):
                                    sel = []
                                    ok_ = True
                                    for q, off in ((2, ok), (1, oj), (0, oi)):
                                        if off >= len(srcs[q]):
                                            ok_ = False
                                            break
                                        sel.append(srcs[q][off])
                                    tm[(ok, oj, oi)] = arr[(Ellipsis,) + tuple(sel)] if ok_ else 0.0
                        return tm
                    A = gather(area)
                    tsum = lambda T: ((T[0, 0, 0] + T[0, 1, 0]) + (T[0, 0, 1] + T[0, 1, 1])) + \
                                     ((T[1, 0, 0] + T[1, 1, 0]) + (T[1, 0, 1] + T[1, 1, 1]))
                    tvol = tsum(A)
                    for mine, theirs in zip(self._flux_sets(self.blocks[b], d), self._flux_sets(f, d)):
                        F = gather(theirs)
                        T = {k: A[k] * F[k] for k in F}
                        mine[(slice(None), dst[2], dst[1], dst[0])] = tsum(T) / tvol

    # ---- time loop (artemis_driver.cpp:145-273; EvolutionDriver upstream, recalled) ------------------------
    def for_each(self, fn):
        for o in self.blocks:
            fn(o)

    def post_init(self):
        self.for_each(lambda o: o.ConsToPrim())
        self.fill_ghosts()
        self.for_each(lambda o: o.PrimToCons())

    def new_dt(self):
        return min(o.new_dt() for o in self.blocks)

    diffusion = False
    gravity = rframe = drag = cooling = False

    def step(self):
        coef = {"rk1": [(0.0, 1.0, 1.0)], "rk2": [(0.0, 1.0, 1.0), (0.5, 0.5, 0.5)],
                "vl2": [(0.0, 1.0, 0.5), (0.0, 1.0, 1.0)],
                "rk3": [(0.0, 1.0, 1.0), (0.25, 0.75, 0.25), (2.0 / 3.0, 1.0 / 3.0, 2.0 / 3.0)]}[self.integrator]
        dust = self.blocks[0].cfg.ns_dust > 0
        self.for_each(lambda o: o.DeepCopyConservedData())
        for stage, (g0, g1, be) in enumerate(coef, 1):
            bdt = be * self.dt
            pcm = stage == 1 and self.integrator == "vl2"
            for o in self.blocks:
                o.CalculateFluxes(GAS, pcm)
                if dust:
                    o.CalculateFluxes(1, pcm)
                if self.diffusion:
                    o.ZeroDiffusionFlux(), o.ViscousFlux(), o.ThermalFlux()
            self.flux_correction()
            for o in self.blocks:
                o.ApplyUpdate(g0, g1, be * self.dt)
                o.FluxSource(bdt, GAS)
                if dust:
                    o.FluxSource(bdt, 1)
                if self.diffusion:
                    o.DiffusionUpdate(bdt)
                if self.gravity:
                    o.ExternalGravity(self.time, bdt)
                if self.rframe:
                    o.RotatingFrameForce(bdt)
                if self.drag:
                    o.DragSource(bdt)
                if self.cooling:
                    o.CoolingSource(self.time, bdt)
                o.SetAuxillaryFields()
                o.ConsToPrim()
            self.fill_ghosts()
            self.for_each(lambda o: o.PrimToCons())

    def evolve(self, tlim=-1.0, nlim=-1):
        if self.dt is None:
            self.dt = self.new_dt()
            if tlim > 0.0 and self.time < tlim and (tlim - self.time) < self.dt:
                self.dt = tlim - self.time
        n = 0
        while (tlim < 0.0 or self.time < tlim) and (nlim < 0 or self.ncycle < nlim):
            self.step()
            self.time += self.dt
            self.ncycle += 1
            n += 1
            dt = self.dt * 2.0 if self.dt < 0.1 * 1.7976931348623157e308 else self.dt
            dt = min(dt, self.new_dt())
            if tlim > 0.0 and self.time < tlim and (tlim - self.time) < dt:
                dt = tlim - self.time
            self.dt = dt
        return n

    def history(self):
        return sum(o.history() for o in self.blocks)

    def block_bounds(self, b):
        c = self.blocks[b].cfg
        return [c.x1min, c.x1max, c.x2min, c.x2max, c.x3min, c.x3max]
