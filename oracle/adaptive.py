"""Adaptive mesh refinement on top of the multilevel CPU oracle.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  An independent restatement -- Python sets and per-block oracles,
no code shared with the product's C++ block tree (artemis_amd/csrc/driver/block_tree.hpp) -- of what happens between
two cycles of an adaptive run of the reference:

  * tagging          Gas::CheckRefinementBlock = ScalarFirstDerivative<FIELD, GEOM> | ScalarMagnitude<FIELD>
                     (gas.cpp:305-380, utils/refinement/amr_criteria.hpp:29-168), the per-block oracle's restatement;
                     task parthenon::Refinement::Tag after EstimateTimestep (artemis_driver.cpp:291-293)
  * flags            upstream, recalled (MeshRefinement::SetRefinement): a block tagged `refine` below the finest
                     level is refined; a block tagged `derefine` above the root level counts consecutive requests and
                     is flagged for derefinement once it has asked `derefine_count` times in a row AND no neighbour
                     (faces, edges, corners) is finer than it; any other tag resets the count
  * new tree         upstream, recalled (Mesh::UpdateMeshBlockTree): refinements first, each restoring 2:1 balance
                     over faces, edges and corners by refining coarser neighbours; then a sibling group is merged
                     when ALL its members are leaves flagged for derefinement and the merge keeps the tree balanced
  * data hand-over   cell-centred fields of unchanged blocks are kept; children of a refined block are
                     ProlongateSharedMinMod<GEOM> of it (prolongation.hpp:83-184; the parent's ghost zones supply the
                     stencil), a merged block is RestrictAverage<GEOM> of its children (restriction.hpp:42-114) --
                     on the conserved variables (Metadata::Independent: gas.cpp:212-240, dust.cpp)
  * afterwards       what Mesh::Initialize does for a modified mesh (upstream, recalled): PreCommFillDerived =
                     ConsToPrim, boundary exchange incl. prolongation and physical conditions, FillDerived =
                     PrimToCons (artemis.cpp:122-123).  SetAuxillaryFields is NOT part of it
                     (fill_derived.cpp:28: "this function is not called during remeshing"); the time step becomes
                     min(2 dt, estimate on the old mesh, estimate on the new mesh) (EvolutionDriver, upstream)
  * initial mesh     Mesh::Initialize's loop: problem generator on every block, first exchange, tag, refine; repeat
                     (the generator fills the new blocks: no prolongation) until the mesh stops changing

Parity unpinned against a Parthenon build (absent submodule; the reference holds no regression answer for its AMR
decks): what this pins is the PRODUCT's remeshing -- tree shape, every leaf's data, dt -- against a second,
independently written implementation of the same published algorithm.
"""
import itertools

from .multilevel import MultiLevelOracle

REFINE, SAME, DEREFINE = 1, 0, -1


class AdaptiveOracle(MultiLevelOracle):
    def __init__(self, mesh_nx, block_nx, xmin, xmax, bc, numlevel, refine_field, refine_type, refine_thr,
                 deref_thr=0.0, derefine_count=10, setup=None, pgen=None, regions=(), **kw):
        """numlevel, derefine_count: <parthenon/mesh>; refine_field ("density" | "pressure"), refine_type
        ("gradient" | "magnitude"), refine_thr, deref_thr: <gas> (gas.cpp:305-380).  setup(block): switch on the
        block's packages (called for every new block oracle and coarse-buffer oracle); pgen(block): the problem
        generator (called with post_init=False semantics: it must only fill the block)."""
        self.max_level = max(0, numlevel - 1)
        self.refine_var = {"density": 0, "pressure": -1}[refine_field]
        self.refine_type, self.refine_thr, self.deref_thr = refine_type, refine_thr, deref_thr
        self.derefine_count = derefine_count
        self.setup, self.pgen = setup or (lambda o: None), pgen
        self.count = {}      # leaf -> consecutive derefinement requests
        self.remeshes = 0
        super().__init__(mesh_nx, block_nx, xmin, xmax, bc, regions=regions, **kw)

    # ---- blocks ------------------------------------------------------------------------------------------
    def make_block(self, level, loc):
        o, c = super().make_block(level, loc)
        self.setup(o), self.setup(c)
        return o, c

    def initialize(self):
        """Mesh::Initialize for a new run: generate, exchange, tag, refine -- until the mesh stops changing."""
        while True:
            for o, c in zip(self.blocks, self.coarse):
                self.pgen(o)
                self.pgen(c)  # (user conditions that re-evaluate the initial profile do so on coarse buffers too)
            self.post_init()
            nb = len(self.leaves)
            changed = self._update_tree(self._flags())
            if changed:
                self.remeshes += 1
                self._build_blocks()     # every block is regenerated by the next pass
                self._carry_counts()
            if len(self.leaves) == nb:
                break
        return self

    # ---- tagging -------------------------------------------------------------------------------------------
    def tags(self):
        out = []
        for o in self.blocks:
            if self.refine_type == "gradient":
                out.append(o.ScalarFirstDerivative(self.refine_var, self.refine_thr)[0])
            else:
                out.append(o.ScalarMagnitude(self.refine_var, self.refine_thr, self.deref_thr)[0])
        return out

    def _levels_around(self, level, loc):
        """Levels of the leaves touching a leaf over faces, edges and corners."""
        out = []
        for o in self.tree.directions():
            kind, _ = self.tree.neighbour(level, loc, o)
            out.append({"same": level, "coarser": level - 1, "finer": level + 1, "phys": level}[kind])
        return out

    def _flags(self):
        """refine_flag of every leaf from this cycle's tags (SetRefinement, upstream, recalled)."""
        flags = {}
        for (level, loc), tag in zip(self.leaves, self.tags()):
            lf = (level, loc)
            if tag >= 0:
                self.count[lf] = 0
            if tag > 0:
                flags[lf] = REFINE if level < self.max_level else SAME
            elif tag < 0:
                if level == 0:
                    flags[lf], self.count[lf] = SAME, 0
                else:
                    self.count[lf] = self.count.get(lf, 0) + 1
                    finer_nb = any(l > level for l in self._levels_around(level, loc))
                    flags[lf] = DEREFINE if (not finer_nb and self.count[lf] >= self.derefine_count) else SAME
            else:
                flags[lf] = SAME
        return flags

    # ---- the tree ------------------------------------------------------------------------------------------
    def _balanced(self, internal):
        """2:1 over faces, edges and corners: every refined node's same-level neighbours exist."""
        t = self.tree
        for level, loc in internal:
            for o in t.directions():
                n = t.wrap(level, tuple(loc[d] + o[d] for d in range(3)))
                if n is None or level == 0:
                    continue
                if (level - 1, t.parent(n)) not in internal:
                    return False
        return True

    def _children(self, level, loc):
        nd = self.ndim
        return [(level + 1, (2 * loc[0] + c1, 2 * loc[1] + c2 if nd > 1 else 0, 2 * loc[2] + c3 if nd > 2 else 0))
                for c3 in range(2 if nd > 2 else 1) for c2 in range(2 if nd > 1 else 1) for c1 in range(2)]

    def _update_tree(self, flags):
        t = self.tree
        before = set(t.internal)
        # refinements (each cascades into coarser neighbours through BlockTree.refine's balance step)
        for (level, loc), f in flags.items():
            if f == REFINE:
                t.refine(level, loc)
        # derefinements: complete sibling groups of leaves, all flagged, whose merge keeps the tree balanced
        parents = {}
        for (level, loc), f in flags.items():
            if f == DEREFINE:
                parents.setdefault((level - 1, t.parent(loc)), []).append((level, loc))
        for par in sorted(parents):
            kids = self._children(*par)
            if par not in t.internal or sorted(parents[par]) != sorted(kids):
                continue
            if any(k in t.internal for k in kids):
                continue
            trial = set(t.internal)
            trial.discard(par)
            if self._balanced(trial) and not self._in_static_region(par):
                t.internal = trial
        return t.internal != before

    def _in_static_region(self, node):
        """A node the <parthenon/static_refinement*> regions keep refined."""
        if not self.regions:
            return False
        probe = type(self.tree)(self.tree.nrb, self.tree.ndim, self.tree.periodic)
        for level, r1, r2, r3 in self.regions:
            probe.add_region(level, (r1[0], r2[0], r3[0]), (r1[1], r2[1], r3[1]), self.xmin, self.xmax)
        return node in probe.internal

    def _carry_counts(self):
        self.count = {lf: self.count.get(lf, 0) for lf in self.leaves}

    # ---- one remesh check after a cycle ------------------------------------------------------------------
    def remesh(self):
        old_leaves = list(self.leaves)
        old = {lf: (o, c) for lf, o, c in zip(self.leaves, self.blocks, self.coarse)}
        if not self._update_tree(self._flags()):
            return False
        self.remeshes += 1
        keep = {lf: old[lf] for lf in self.tree.leaves() if lf in old}
        for lf, (o, _) in old.items():  # back-reaction sums of blocks that leave the mesh stay in the total
            if lf not in keep and hasattr(o, "_npart"):
                f = o.nbody_force()
                self._force_retired = f if self._force_retired is None else self._force_retired + f
        self._build_blocks(keep)
        self._carry_counts()
        S = [self._start(d) for d in range(3)]
        n = self.nx
        half = [n[d] // 2 if d < self.ndim else 1 for d in range(3)]
        fields = ["gas.cons"] + (["dust.cons"] if self.has_dust else [])
        for lf, o, cbuf in zip(self.leaves, self.blocks, self.coarse):
            if lf in old:
                continue
            # a new block first runs the problem generator like every block of a new mesh: conditions that re-evaluate
            # the initial profile (the disk problem's `ic`, pgen/disk.hpp:597-632) need it on the block's own zone
            # centres and on its coarse buffer's; the state itself is overwritten by the hand-over below
            if self.pgen is not None:
                self.pgen(o), self.pgen(cbuf)
            level, loc = lf
            par = (level - 1, self.tree.parent(loc)) if level > 0 else None
            if par in old:  # a child of a refined block: prolongate its octant of the parent
                c = [loc[d] & 1 if d < self.ndim else 0 for d in range(3)]
                lo = [S[d] + c[d] * half[d] for d in range(3)]
                hi = [lo[d] + half[d] - 1 if d < self.ndim else lo[d] for d in range(3)]
                rng = (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
                for f in fields:
                    o.ProlongateSharedMinMod(old[par][0], rng, lo, S, field=f)
            else:           # a merged block: restrict every child into its octant
                for kid in self._children(level, loc):
                    c = [kid[1][d] & 1 if d < self.ndim else 0 for d in range(3)]
                    lo = [S[d] + c[d] * half[d] for d in range(3)]
                    hi = [lo[d] + half[d] - 1 if d < self.ndim else lo[d] for d in range(3)]
                    rng = (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
                    for f in fields:
                        old[kid][0].RestrictAverage(o, rng, lo, S, field=f)
        assert all(lf in old or (lf[0] > 0 and (lf[0] - 1, self.tree.parent(lf[1])) in old) or
                   all(k in old for k in self._children(*lf)) for lf in self.leaves), "remesh by more than one level"
        del old_leaves
        # Mesh::Initialize for a modified mesh: ConsToPrim -> exchange -> PrimToCons (no SetAuxillaryFields)
        self.post_init()
        return True

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
            est = self.new_dt()            # PostStepTasks: EstimateTimestep on the mesh the step ran on
            if self.remesh():              # LoadBalancingAndAdaptiveMeshRefinement, then the estimate on the new mesh
                est = min(est, self.new_dt())
            dt = self.dt * 2.0 if self.dt < 0.1 * 1.7976931348623157e308 else self.dt
            dt = min(dt, est)
            if tlim > 0.0 and self.time < tlim and (tlim - self.time) < dt:
                dt = tlim - self.time
            self.dt = dt
        return n

    _force_retired = None

    def nbody_force(self):
        """The particle_force rows accumulated over the run (nbody_gravity.hpp:210-215), summed over all blocks that
        ever existed."""
        tot = sum(o.nbody_force() for o in self.blocks)
        return tot if self._force_retired is None else tot + self._force_retired

    def level_counts(self):
        out = {}
        for level, _ in self.leaves:
            out[level] = out.get(level, 0) + 1
        return out
