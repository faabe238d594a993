// ColumnGrid with the call surface of the reference class (template/script_columngrid.js:12-43
// and the methods the runtime and the consumers use), running on the MI355X through the
// colgrid_napi addon.  It can be put in the slot the reference publishes its class in
// (window.prj.ColumnGrid, template/script_columngrid3.js:321-323): same constructor, same
// method names and arity, synchronous, `undefined` results, silent no-op on bad thickness.
//
// What is different inside:
//  * the grid lives in GPU memory; `cols` is materialised (device -> host, reference layout)
//    only when somebody reads it, and dropped again by the next step;
//  * a mask is rasterised once per step on the device.  deposit_general / etch_general accept
//    the reference's closure maskFn(x, y); a closure made by maskFunctionFor() carries the mask
//    record's shape list and is rasterised by the device kernel, any other closure is evaluated
//    here at every cell centre (cg1:448) into a byte raster and uploaded;
//  * deposit_ALD / etch_wet keep their per-increment anchors (sliderCache.aldCache /
//    sliderCache.wetetchCache, cg2:131-136, 911) as device snapshots and the ALD seeds in the
//    handle; replacing `sliderCache` (script_runtime.js:275) releases them;
//  * saveCache keeps device snapshots (with their Lmax -- the reference forgets it, cg1:229-235
//    vs 276-279) and the same [snapshot, useCount] bookkeeping the runtime pokes at.
'use strict';
const path = require('path');
const addon = require(path.join(__dirname, 'colgrid_napi.node'));
const { reduceMask, packShapes } = require('./maskshapes.js');

const MASK_NONE = -1;
const SHAPES = Symbol.for('colgrid.maskShapes');

// A closure with the reference's predicate semantics (script_runtime.js:54-71) that also
// carries the reduced shape list, so that the grid can rasterise it on the device.
function maskFunctionFor(maskData, jsPredicate) {
  const reduced = reduceMask(maskData);
  const fn = jsPredicate || (() => { throw new Error('device-only mask closure called on the host'); });
  fn[SHAPES] = reduced;
  return fn;
}

class ColumnGrid {
  // opts.devices (no counterpart in the reference): HIP device ordinals, one per slab.  The grid is then cut into that
  // many slabs of rows, slab k on device devices[k] (repeats allowed), and the library drives them as one grid
  // (cg_group_*, include/colgrid.h): every method below behaves as on one device, results included.
  constructor(LXnm, LYNm, dx, dy, LmaxInit = 32, zScale = 0.1, opts = {}) {
    this.zScale = zScale;
    this.expandStep = 16;
    this.nMaxCache = 10;
    const devices = opts && opts.devices;
    this._h = (devices && devices.length > 0)
      ? addon.createGroup(Number(LXnm), Number(LYNm), Number(dx), Number(dy), LmaxInit, zScale, Int32Array.from(devices))
      : addon.create(Number(LXnm), Number(LYNm), Number(dx), Number(dy), LmaxInit, zScale);
    this._cols = null;
    this._maskIds = new Map();          // reduced shape list (the object a mask closure carries) -> device mask id
    this._nextSlot = 0;
    this.colsCache = {};
    this._sliderCache = {};
    this.keepStepSnapshots = true;      // the reference stores a grid copy per ALD / wet-etch increment
    this._refreshInfo();
  }

  // sliderCache is a plain object for its users; the device snapshots and seeds it stands for
  // are released when it is replaced (cg1:244, script_runtime.js:275).
  get sliderCache() { return this._sliderCache; }
  set sliderCache(v) {
    const old = this._sliderCache;
    if (old && old !== v && this._h) {
      if (old.wetetchCache) for (const e of old.wetetchCache.grid.values()) addon.snapshotDrop(this._h, e.slot);
      if (old.aldCache) {
        for (const e of old.aldCache.gridWithCavity.values()) addon.snapshotDrop(this._h, e.slot);
        addon.aldDropSeeds(this._h);
      }
    }
    this._sliderCache = v || {};
  }

  _refreshInfo() {
    const i = addon.info(this._h);
    this.NX = i.nx; this.NY = i.ny; this.Ncols = i.nx * i.ny;
    this.dx = i.dx; this.dy = i.dy;
    this.offsetX = i.offsetX; this.offsetY = i.offsetY; this.LXeff = i.lxEff; this.LYeff = i.lyEff;
    this.Lmax = i.lmax;
  }

  dispose() { if (this._h) { addon.destroy(this._h); this._h = null; } }

  // ---- the `cols` contract of the consumers (script_gridrenderer.js:216-263)
  get cols() {
    if (!this._cols) {
      const d = addon.download(this._h);
      this._cols = { mat: d.mat, zpair: d.zpair, len: d.len };
      this.Lmax = d.lmax;
    }
    return this._cols;
  }
  set cols(v) {                                  // loadCache-style replacement by host arrays
    addon.upload(this._h, v.mat, v.zpair, v.len, (v.mat.length / Math.max(1, v.len.length)) | 0);
    this._touched();
  }
  _touched() { this._cols = null; this.Lmax = addon.info(this._h).lmax; }

  _quantizeZ(zReal) { return Math.round(zReal / this.zScale); }     // cg1:117-119
  _dequantizeZ(zInt) { return zInt * this.zScale; }                 // cg1:120-122
  _colIndex(i, j) { return i * this.NY + j; }
  _segIndex(i, j, k) { return (i * this.NY + j) * this.Lmax + k; }

  // ---- masks
  _maskId(maskFn) {
    if (!maskFn) return MASK_NONE;                                   // "no mask" (cg1:449)
    const reduced = maskFn[SHAPES];
    if (reduced) {
      if (reduced.open) return MASK_NONE;
      // keyed by the identity of the reduced list: a closure that is handed in again (the runtime keeps one per mask
      // record) costs a map lookup, not a walk over its shapes
      let id = this._maskIds.get(reduced);
      if (id === undefined) {
        const p = packShapes(reduced);
        id = addon.maskFromShapes(this._h, p.shapes, p.nshapes, p.pts);
        this._maskIds.set(reduced, id);
        if (this._maskIds.size > 64) {                               // bound the device rasters we keep
          const [k0, id0] = this._maskIds.entries().next().value;
          addon.maskDestroy(this._h, id0); this._maskIds.delete(k0);
        }
      }
      return id;
    }
    // an arbitrary closure: evaluate it at every cell centre, as the reference does (cg1:446-449)
    const raster = new Uint8Array(this.Ncols);
    for (let i = 0; i < this.NX; i++) {
      const x = (i + 0.5) * this.dx;
      for (let j = 0; j < this.NY; j++) raster[i * this.NY + j] = maskFn(x, (j + 0.5) * this.dy) ? 1 : 0;
    }
    return { temp: addon.maskFromRaster(this._h, raster) };
  }
  _withMask(maskFn, body) {
    const m = this._maskId(maskFn);
    const id = (typeof m === 'object') ? m.temp : m;
    try { body(id); } finally { if (typeof m === 'object') addon.maskDestroy(this._h, id); }
    this._touched();
  }

  // ---- process steps (same names and argument order as the reference)
  deposit_general(maskFn, matId, thickness, conformality) {          // cg1:435
    const t = Math.max(0, Number(thickness) || 0);
    if (t <= 0) return;
    const S = Number(conformality === undefined || conformality === null ? 1.0 : conformality);
    this._withMask(maskFn, id => addon.depositGeneral(this._h, id, matId | 0, t, S));
  }
  etch_general(maskFn, matId, thickness, conformality) {             // cg1:598
    const t = Math.max(0, Number(thickness) || 0);
    if (t <= 0) return;
    const S = Number(conformality === undefined || conformality === null ? 1.0 : conformality);
    this._withMask(maskFn, id => addon.etchGeneral(this._h, id, matId | 0, t, S));
  }
  cmp(depth, stopMatId = 0) { addon.cmp(this._h, Number(depth), stopMatId | 0); this._touched(); }       // cg1:747
  strip_connected(matId) { addon.stripConnected(this._h, matId | 0); this._touched(); }                  // cg1:867
  identify_cavity() { addon.identifyCavity(this._h); this._touched(); }                                  // cg2:9
  maxHeight() { return addon.maxHeight(this._h); }                                                       // cg2:85

  _anchorAtOrBelow(map, T) {                                         // cg2:832-835, cg3:274-277
    let best = -1;
    for (const k of map.keys()) if (k <= T + 1e-9 && k > best) best = k;
    return best;
  }
  _saveAnchor(map, key) {
    if (!this.keepStepSnapshots) return;
    let e = map.get(key);
    if (!e) { e = { slot: this._nextSlot++ }; map.set(key, e); }
    addon.snapshotSave(this._h, e.slot);
  }

  // deposit_ALD(matId, totalThk, { dt, isCache }) (cg2:111-868)
  deposit_ALD(matId, totalThk, opts = {}) {
    const T = Math.max(0, Number(totalThk) || 0);
    if (T <= 0) return;
    const zStepNm = this._zStepNm || Math.min(this.dx, this.dy);
    const dt = Math.max(opts.dt === undefined || opts.dt === null ? Math.max(this.dx, this.dy, zStepNm) : opts.dt, 1e-3);
    const isCache = !!opts.isCache;
    const sc = this._sliderCache;
    sc.aldCache = sc.aldCache || { exposedSegments: null, gridWithCavity: new Map(), BitsCache: {} };
    const ac = sc.aldCache;
    try {
      if (!(isCache && ac.exposedSegments)) ac.exposedSegments = { count: addon.aldCollectSeeds(this._h, T) };
      let tAnchor = 0;
      if (isCache && ac.gridWithCavity.size > 0) {
        const best = this._anchorAtOrBelow(ac.gridWithCavity, T);
        if (best >= 0) { addon.snapshotLoad(this._h, ac.gridWithCavity.get(best).slot); tAnchor = best; }
      }
      for (let t = tAnchor + dt; t < T - 1e-9; t += dt) {
        addon.aldIncrement(this._h, matId | 0, t);
        this._saveAnchor(ac.gridWithCavity, t);
      }
      addon.aldCoat(this._h, matId | 0, T);
      this._saveAnchor(ac.gridWithCavity, T);
      addon.identifyCavity(this._h);
    } finally {
      addon.releaseScratch(this._h);
      this._touched();
    }
  }

  // etch_wet(matId, totalThk, { dt, isCache }) (cg2:893 - cg3:308)
  etch_wet(matId, totalThk, opts = {}) {
    const T = Math.max(0, Number(totalThk) || 0);
    if (T <= 0) return;
    const baseDt = Math.max(opts.dt === undefined || opts.dt === null ? Math.min(this.dx, this.dy) : opts.dt, 1e-6);
    const isCache = !!opts.isCache;
    const INV_SQRT2 = 0.7071067811865476;
    const sc = this._sliderCache;
    sc.wetetchCache = sc.wetetchCache || { grid: new Map(), R: 0 };
    const wc = sc.wetetchCache;
    let R = 0, t = 0;
    try {
      if (isCache && wc.grid.size > 0) {
        const best = this._anchorAtOrBelow(wc.grid, T);
        if (best >= 0) { addon.snapshotLoad(this._h, wc.grid.get(best).slot); R = Number(wc.R) || 0; t = best; }
      }
      while (t < T - 1e-9) {
        R += INV_SQRT2;
        let useDiagonal = false;
        if (R > 1) { R -= 1; useDiagonal = true; }
        const dtQ = Math.max(1, this._quantizeZ(Math.min(baseDt, T - t)) | 0);
        addon.etchWetStep(this._h, matId | 0, dtQ, useDiagonal ? 1 : 0);
        t += this._dequantizeZ(dtQ);
        this._saveAnchor(wc.grid, t);
        wc.R = R;
      }
      addon.identifyCavity(this._h);
    } finally {
      addon.releaseScratch(this._h);
      this._touched();
    }
  }

  // ---- grid lifetime
  createNewGrid(LmaxInit = this.Lmax) { addon.reset(this._h, LmaxInit | 0); this._touched(); }            // cg1:40
  clearAll() { addon.clear(this._h); this._touched(); }                                                   // cg1:223
  cloneCols() {                                                                                           // cg1:229
    const c = this.cols;
    return { mat: new Uint8Array(c.mat), zpair: new Uint16Array(c.zpair), len: new Uint8Array(c.len) };
  }

  // ---- accessors kept for API parity (cg1:174-218)
  addLayer(i, j, matId, z0_real, z1_real) {                                                               // cg1:196
    addon.addLayer(this._h, i | 0, j | 0, matId & 255, Number(z0_real), Number(z1_real)); this._touched();
  }
  removeTopLayer(i, j) { addon.removeTopLayer(this._h, i | 0, j | 0); this._touched(); }                  // cg1:213
  setPlaneBudget(maxPlanes) { addon.setPlaneBudget(this._h, maxPlanes | 0); }   // engine knob, no counterpart in the reference
  _nextPowerOf2(n) { return 1 << (32 - Math.clz32(Math.max(1, n - 1))); }                                 // cg1:302
  _clamp(v, lo, hi) { return Math.max(lo, Math.min(hi, v)); }                                             // cg1:309
  getLen(i, j) { return this.cols.len[this._colIndex(i, j)]; }
  topMat(i, j) { const c = this._colIndex(i, j), l = this.cols.len[c]; return l === 0 ? 0 : this.cols.mat[c * this.Lmax + l - 1]; }
  topZ(i, j) { const c = this._colIndex(i, j), l = this.cols.len[c]; return l === 0 ? 0 : this._dequantizeZ(this.cols.zpair[(c * this.Lmax + l - 1) * 2 + 1]); }

  // ---- step cache (cg1:241-288): device snapshots, host-side bookkeeping as in the reference
  initializeCache() {
    for (const k of Object.keys(this.colsCache)) this._dropSnap(this.colsCache[k]);
    this.colsCache = {};
    this.sliderCache = {};                       // through the setter: releases the step anchors
  }
  _dropSnap(entry) { if (entry && entry[0] && entry[0].slot !== undefined) addon.snapshotDrop(this._h, entry[0].slot); }
  saveCache(id) {
    while (Object.keys(this.colsCache).length >= this.nMaxCache) {
      const cand = Object.entries(this.colsCache).map(([key, [, num]]) => ({ key, num }));
      if (cand.length === 0) break;
      const min = cand.reduce((a, b) => (a.num < b.num ? a : b));       // least used, first among equals
      this._dropSnap(this.colsCache[min.key]);
      delete this.colsCache[min.key];
    }
    if (this.colsCache[id]) this._dropSnap(this.colsCache[id]);
    const slot = this._nextSlot++;
    addon.snapshotSave(this._h, slot);
    this.colsCache[id] = [{ slot }, 0];
  }
  loadCache(id) {
    if (!id) { this.createNewGrid(); return; }
    const c = this.colsCache[id];
    if (!c[0]) { this.clearAll(); return; }
    addon.snapshotLoad(this._h, c[0].slot);
    this._touched();
    this.colsCache[id][1] += 1;
  }
  clearCacheExcept(ids) {
    Object.keys(this.colsCache).forEach(k => {
      if (!ids.includes(k)) { this._dropSnap(this.colsCache[k]); delete this.colsCache[k]; }
    });
  }

  // ---- cross-sections (what template/script_slice2d.js:15-40 reads, on the real grid)
  indexFromX(xnm) { return addon.indexFromPos(this._h, 0, Number(xnm)); }
  indexFromY(ynm) { return addon.indexFromPos(this._h, 1, Number(ynm)); }
  worldX(i) { return this.offsetX + i * this.dx; }
  worldY(j) { return this.offsetY + j * this.dy; }
  // axis 'YZ' (fixed i) or 'XZ' (fixed j); returns { mat, zpair, len, lmax, line } for the requested lines
  slices(axis, indices) { return addon.slices(this._h, axis === 'XZ' ? 1 : 0, Int32Array.from(indices)); }

  checksum() { return addon.checksum(this._h); }
  // { rccl, exchanges, bytes }: how rows have crossed slab edges so far (all zero on one device)
  transport() { return addon.groupTransport(this._h); }
}

module.exports = { ColumnGrid, maskFunctionFor, addon };
