// TEST INFRASTRUCTURE ONLY -- generates tests/golden/* by running the reference's own
// JavaScript (read in place from /root/reference, see ref_harness.js) on the inputs below.
//
//   node --harmony-nullish --harmony-optional-chaining oracle/gen_golden.js
//
// Each case becomes
//   tests/golden/<name>.json      inputs (domain, fabric mask JSON, reduced shapes, steps) and
//                                 per-step FNV hashes / Lmax of the reference's grid
//   tests/golden/<name>.bin.gz    full dumps: for every step marked dump (and the last one)
//                                 len[N] | mat[N*Lmax] | zpair[2*N*Lmax] little endian,
//                                 in the order of the "dumps" list of the JSON
//   tests/golden/<name>.masks.gz  one byte per column (1 = open) for every mask, in the order
//                                 of the "mask_order" list, produced by the reference's
//                                 _isPointBlocked
// Only data (inputs and outputs) is written; no reference source text.
'use strict';
const fs = require('fs');
const path = require('path');
const zlib = require('zlib');
const H = require('./ref_harness.js');
const { reduceMask } = require('../fem_amd/js/maskshapes.js');

const OUT = path.join(__dirname, '..', 'tests', 'golden');
fs.mkdirSync(OUT, { recursive: true });
const ref = H.loadReference();

// ---- fabric-style object builders (field names of fabric 5.3.0 toJSON as the reference reads them)
const rect = (left, top, width, height, o = {}) => Object.assign(
  { type: 'rect', left, top, width, height, angle: 0, scaleX: 1, scaleY: 1, visible: true,
    data: { polarity: 'positive' } }, o);
const circle = (left, top, radius, o = {}) => Object.assign(
  { type: 'circle', left, top, width: 2 * radius, height: 2 * radius, radius, angle: 0, scaleX: 1, scaleY: 1,
    visible: true, data: { polarity: 'positive' } }, o);
const ellipse = (left, top, rx, ry, o = {}) => Object.assign(
  { type: 'ellipse', left, top, width: 2 * rx, height: 2 * ry, rx, ry, angle: 0, scaleX: 1, scaleY: 1,
    visible: true, data: { polarity: 'positive' } }, o);
const polygon = (left, top, points, o = {}) => Object.assign(
  { type: 'polygon', left, top, width: 0, height: 0, points, angle: 0, scaleX: 1, scaleY: 1,
    visible: true, data: { polarity: 'positive' } }, o);
const neg = { data: { polarity: 'negative' } };

function periodicRects(L, pitch, size, shift = 0) {
  const objs = [];
  for (let x = shift; x < L; x += pitch) for (let y = shift; y < L; y += pitch) objs.push(rect(x, y, size, size));
  return { objects: objs };
}

function maskFunction(spec, masks) {
  if (spec === null || spec === undefined) return null;
  if (spec === 'all') return () => true;
  return H.maskPredicate(ref, masks[spec]);
}

function dumpGrid(g) {
  const { mat, zpair, len } = g.cols;
  return Buffer.concat([Buffer.from(len.buffer, len.byteOffset, len.byteLength),
    Buffer.from(mat.buffer, mat.byteOffset, mat.byteLength),
    Buffer.from(zpair.buffer, zpair.byteOffset, zpair.byteLength)]);
}

function runCase(name, domain, masks, steps, opts = {}) {
  const g = new ref.ColumnGrid(domain.LX, domain.LY, domain.dx, domain.dy,
    domain.Lmax === undefined ? 32 : domain.Lmax, domain.zScale === undefined ? 0.1 : domain.zScale);
  const out = { name, domain, nx: g.NX, ny: g.NY, masks, mask_order: Object.keys(masks), reduced: {},
    steps: [], dumps: [], lmax_init: g.Lmax };
  // rasters of every mask by the reference predicate
  const rasters = [];
  for (const m of out.mask_order) {
    const fn = H.maskPredicate(ref, masks[m]);
    const r = Buffer.alloc(g.NX * g.NY);
    for (let i = 0; i < g.NX; i++) for (let j = 0; j < g.NY; j++)
      r[i * g.NY + j] = fn((i + 0.5) * g.dx, (j + 0.5) * g.dy) ? 1 : 0;
    rasters.push(r);
    out.reduced[m] = reduceMask(masks[m]);
  }
  const bins = [];
  steps.forEach((s, idx) => {
    const fn = maskFunction(s.mask, masks);
    switch (s.op) {
      case 'deposit_general': g.deposit_general(fn, s.mat, s.t, s.S); break;
      case 'etch_general': g.etch_general(fn, s.mat, s.t, s.S); break;
      case 'cmp': g.cmp(s.t, s.mat); break;
      case 'strip_connected': g.strip_connected(s.mat); break;
      case 'identify_cavity': g.identify_cavity(); break;
      case 'reset': g.createNewGrid(); break;
      case 'etch_wet': g.etch_wet(s.mat, s.t, s.dt === undefined ? { isCache: !!s.isCache } : { isCache: !!s.isCache, dt: s.dt }); break;
      case 'deposit_ALD': g.deposit_ALD(s.mat, s.t, s.dt === undefined ? { isCache: !!s.isCache } : { isCache: !!s.isCache, dt: s.dt }); break;
      case 'slider_reset': g.sliderCache = {}; break;
      case 'addLayer': g.addLayer(s.i, s.j, s.mat, s.z0, s.z1); break;
      case 'removeTopLayer': g.removeTopLayer(s.i, s.j); break;
      default: throw new Error('unknown op ' + s.op);
    }
    const rec = Object.assign({}, s, { fnv: H.gridHash(g), lmax: g.Lmax, max_height: g.maxHeight() });
    if (s.i !== undefined) rec.probe = [g.getLen(s.i, s.j), g.topMat(s.i, s.j), g.topZ(s.i, s.j)];
    if (g.sliderCache && (g.sliderCache.aldCache || g.sliderCache.wetetchCache)) {
      const a = g.sliderCache.aldCache, w = g.sliderCache.wetetchCache;
      rec.slider = { ald_keys: a ? Array.from(a.gridWithCavity.keys()) : [], seeds: a && a.exposedSegments ? a.exposedSegments.length : 0,
        wet_keys: w ? Array.from(w.grid.keys()) : [], wet_R: w ? w.R : 0 };
    }
    if (s.dump || idx === steps.length - 1) {
      if (!opts.noDump) { rec.dump_index = out.dumps.length; out.dumps.push({ step: idx, lmax: g.Lmax }); bins.push(dumpGrid(g)); }
    }
    out.steps.push(rec);
  });
  fs.writeFileSync(path.join(OUT, name + '.json'), JSON.stringify(out));
  if (bins.length) fs.writeFileSync(path.join(OUT, name + '.bin.gz'), zlib.gzipSync(Buffer.concat(bins), { level: 9 }));
  if (rasters.length && !opts.noRaster)
    fs.writeFileSync(path.join(OUT, name + '.masks.gz'), zlib.gzipSync(Buffer.concat(rasters), { level: 9 }));
  else if (rasters.length) {
    // large grid: keep only an FNV per raster
    out.raster_fnv = rasters.map(r => { let h = 2166136261 >>> 0; for (const b of r) h = Math.imul(h ^ b, 16777619) >>> 0; return ('00000000' + h.toString(16)).slice(-8); });
    fs.writeFileSync(path.join(OUT, name + '.json'), JSON.stringify(out));
  }
  console.log(name, g.NX + 'x' + g.NY, 'Lmax', g.Lmax, 'final', out.steps[out.steps.length - 1].fnv);
}

const D = (LX, LY, dx = 2, dy = 2) => ({ LX, LY, dx, dy });
const dep = (mask, mat, t, S = 0, dump = false) => ({ op: 'deposit_general', mask, mat, t, S, dump });
const etc = (mask, mat, t, S = 0, dump = false) => ({ op: 'etch_general', mask, mat, t, S, dump });

// ---------------------------------------------------------------------------------------------
// 1. vertical paths: extend / append / partial and whole-layer removal / ALL / emptying columns
{
  const masks = {
    sq: { objects: [rect(0, 0, 200, 200), rect(20, 24, 40, 48, neg)] },     // open only inside the inner rect
    stripes: { objects: [rect(0, 0, 10, 200), rect(30, 0, 14, 200), rect(60, 0, 6, 200)] },
    none: { objects: [] },
  };
  runCase('vertical_small', D(80, 112), masks, [
    dep('all', 3, 30, 0, true), dep('all', 3, 2.5, 0), dep('all', 4, 20, 0, true),
    dep('stripes', 5, 7.25, 0, true), etc('sq', 5, 3.1, 0, true), etc('sq', 4, 50, 0, true),
    etc('stripes', 255, 26.05, 0, true), dep(null, 6, 0.04, 0), dep(null, 6, 0.05, 0, true),
    etc('none', 6, 0.05, 0), etc(null, 3, 5, 0), etc(null, 255, 1000, 0, true),
    dep('sq', 4, 12, 0.1), dep('all', 7, 1.95, 1.0), etc('all', 7, 1.0, 1.0),
  ]);
}

// 2. conformal deposit + isotropic etch (the lateral stencil), several radii / conformalities
{
  const masks = {
    trench: { objects: [rect(0, 0, 300, 300), rect(30, 0, 16, 300, neg), rect(70, 20, 24, 60, neg)] },
    island: { objects: [rect(0, 0, 300, 300), circle(40, 30, 18, neg)] },
    half: { objects: [rect(0, 0, 60, 300)] },
  };
  runCase('stencil_small', D(128, 104), masks, [
    dep('all', 3, 30, 0), dep('all', 4, 20, 0), etc('trench', 4, 20, 0), etc('trench', 3, 12, 0, true),
    dep('all', 5, 10, 1.0, true), etc(null, 5, 6, 1.0, true), dep('half', 6, 7.3, 0.5, true),
    etc('island', 255, 9, 0.7, true), dep(null, 4, 4, 1.0, true), etc('all', 4, 14, 1.0, true),
    etc('half', 5, 6.5, 0.4, true), dep('island', 7, 12.7, 0.85, true), etc(null, 255, 3, 1.0),
  ]);
}

// 3. the same ops on a non-square grid with odd sizes and a big radius (R = 12)
{
  const masks = { holes: { objects: [circle(10, 14, 9), circle(50, 40, 13), ellipse(20, 70, 22, 8), rect(80, 10, 12, 90, { angle: 20 })] } };
  runCase('stencil_odd', D(134, 90), masks, [
    dep(null, 3, 40, 0), etc('holes', 3, 22.2, 0, true), dep(null, 4, 24, 1.0, true), etc(null, 4, 11, 0.6, true),
    etc('holes', 255, 7.7, 1.0, true), dep('holes', 5, 3, 1.0),
  ]);
}

// 4. the no-gap anchor sequence on 256^2 (hashes only)
{
  const N = 256;
  const masks = { sq: { objects: [rect(0, 0, 2 * N, 2 * N), rect(0.5 * N, 0.5 * N, N, N, neg)] } };
  runCase('anchor_256', D(512, 512), masks, [
    dep('all', 3, 30, 0), dep('all', 4, 20, 0), etc('sq', 4, 20, 0), dep('all', 5, 10, 1.0), etc(null, 5, 6, 1.0),
  ], { noDump: true, noRaster: true });
}

// 5. mask shapes: every type, rotation, scaling, polarity stacking, invisible, path, unknown type
{
  const tri = [{ x: 0, y: 0 }, { x: 40, y: 6 }, { x: 14, y: 44 }];
  const star = [{ x: 20, y: 0 }, { x: 26, y: 14 }, { x: 40, y: 14 }, { x: 29, y: 23 }, { x: 33, y: 38 }, { x: 20, y: 29 }, { x: 7, y: 38 }, { x: 11, y: 23 }, { x: 0, y: 14 }, { x: 14, y: 14 }];
  const masks = {
    rects: { objects: [rect(10, 10, 30, 20), rect(50, 40, 25, 25, { angle: 30 }), rect(100, 20, 10, 10, { scaleX: 2.5, scaleY: 0.8, angle: -15 })] },
    circles: { objects: [circle(20, 20, 15), circle(70, 30, 10, { scaleX: 1.7 }), ellipse(40, 80, 30, 12, { angle: 45 }), circle(100, 90, 0, { width: 24, height: 24, radius: 0 })] },
    polys: { objects: [polygon(12, 8, tri), polygon(70, 50, star, { angle: 10, scaleX: 1.2, scaleY: 1.1 }),
      { type: 'path', left: 20, top: 80, width: 0, height: 0, angle: 0, scaleX: 1, scaleY: 1, visible: true, data: {},
        path: [['M', 0, 0], ['L', 30, 0], ['Q', 35, 10, 30, 20], ['L', 0, 20], ['z']] }] },
    stack: { objects: [rect(0, 0, 160, 140), circle(30, 30, 25, neg), rect(40, 40, 12, 12), rect(90, 20, 40, 90, neg),
      rect(100, 30, 20, 20, { visible: false }), { type: 'triangle', left: 0, top: 0, width: 300, height: 300, visible: true, angle: 0, scaleX: 1, scaleY: 1 },
      rect(95, 60, 30, 10, { angle: 33.3 })] },
    nodata: { objects: [{ type: 'rect', left: 30, top: 30, width: 50, height: 40, visible: true }, { type: 'rect', left: 0, top: 0, width: 400, height: 5 }] },
    c1: { objects: [rect(100 / 4, 100 / 4, 200 / 4, 120 / 4), circle(300 / 4, 300 / 4, 60 / 4), rect(140 / 4, 130 / 4, 40 / 4, 40 / 4, Object.assign({ angle: 30 }, neg))] },
  };
  const steps = [dep('all', 3, 10, 0)];
  for (const m of Object.keys(masks)) steps.push(dep(m, 4 + steps.length, 5, 0, true));
  runCase('masks_small', D(160, 140), masks, steps);
}

// 6. Lmax growth: more than 32 layers (32 -> 48), with a masked region left behind
{
  const masks = { left: { objects: [rect(20, 0, 100, 100)] } };
  const steps = [];
  for (let k = 0; k < 36; k++) steps.push(dep(k % 3 === 2 ? 'left' : 'all', 3 + (k % 2), 1.5 + 0.1 * k, 0, k === 31 || k === 32));
  steps.push(etc('left', 255, 9.3, 0, true));
  steps.push(dep(null, 9, 5, 1.0));
  runCase('growth_small', D(40, 24), masks, steps);
}

// 7. general cavity handling: strip a buried layer at a trench wall -> gaps -> identify_cavity
{
  const masks = {
    trench: { objects: [rect(0, 0, 200, 200), rect(24, 10, 14, 50, neg)] },
    pit: { objects: [rect(0, 0, 200, 200), rect(60, 30, 10, 10, neg)] },
  };
  runCase('cavity_small', D(96, 72), masks, [
    dep('all', 3, 20, 0), dep('all', 4, 8, 0), dep('all', 5, 12, 0), etc('trench', 255, 25, 0, true),
    { op: 'strip_connected', mat: 4, dump: true }, { op: 'identify_cavity', dump: true },
    dep('all', 6, 6, 1.0, true), etc('pit', 255, 30, 0), { op: 'strip_connected', mat: 4, dump: true },
    dep(null, 7, 5, 1.0, true), { op: 'cmp', t: 4, mat: 0, dump: true }, { op: 'cmp', t: 30, mat: 5, dump: true },
    { op: 'cmp', t: 30, mat: 3, dump: true }, etc('all', 255, 3, 1.0, true),
  ]);
}

// 7a. a lateral deposit whose mask opens no cell returns before identify_cavity (cg1:474-485): a cavity that an
//     etch has opened to the air stays until the next cavity pass; vertical deposits and etches with the same mask
{
  const masks = {
    trench: { objects: [rect(0, 0, 200, 200), rect(24, 10, 14, 50, neg)] },
    pit: { objects: [rect(0, 0, 200, 200), rect(30, 20, 6, 6, neg)] },
    closed: { objects: [rect(-10, -10, 400, 400)] },
  };
  runCase('empty_mask_small', D(96, 72), masks, [
    dep('all', 3, 20, 0), dep('all', 4, 8, 0), dep('all', 5, 12, 0), etc('trench', 255, 25, 0),
    { op: 'strip_connected', mat: 4 }, dep('all', 6, 6, 1.0, true), etc('pit', 255, 14, 0, true),
    dep('closed', 7, 6, 1.0, true), dep('closed', 7, 6, 0, true), etc('closed', 255, 5, 1.0, true),
    dep(null, 7, 3, 1.0, true),
  ]);
}

// 7b. wet etch and ALD (cg2:111 - cg3:308): undercut, pinch-off cavities, the slider caches,
//     and thin gaps that put several runs of air into one 32-bin word of the ALD bitsets
{
  const wet = (mat, t, o = {}) => Object.assign({ op: 'etch_wet', mat, t, dump: true }, o);
  const ald = (mat, t, o = {}) => Object.assign({ op: 'deposit_ALD', mat, t, dump: true }, o);
  const masks = {
    trench: { objects: [rect(0, 0, 200, 200), rect(24, 8, 14, 30, neg), rect(46, 20, 6, 20, neg)] },
    pit: { objects: [rect(0, 0, 200, 200), rect(8, 30, 8, 8, neg)] },
  };
  runCase('wet_ald_small', D(64, 48), masks, [
    dep('all', 3, 20, 0), dep('all', 4, 8, 0), dep('all', 5, 12, 0), etc('trench', 255, 25, 0, true),
    wet(4, 3), ald(6, 3), wet(255, 2.5), ald(7, 5), etc('pit', 255, 60, 0), wet(3, 5, { dt: 1.5 }), ald(4, 4.5, { dt: 1 }),
    { op: 'cmp', t: 6, mat: 0, dump: true }, wet(7, 2),
  ]);
  runCase('wet_ald_cache', D(48, 40), masks, [
    dep('all', 3, 20, 0), dep('all', 4, 8, 0), dep('all', 5, 12, 0), etc('trench', 255, 25, 0, true),
    wet(4, 6), wet(4, 4, { isCache: true }), wet(4, 7.5, { isCache: true }), wet(4, 1, { isCache: true }),
    { op: 'slider_reset' }, ald(6, 6), ald(6, 3, { isCache: true }), ald(6, 7, { isCache: true }), ald(6, 1.5, { isCache: true }),
    { op: 'slider_reset' }, wet(255, 3.3, { isCache: true }), ald(7, 2.2, { isCache: true }),
  ]);
  runCase('wet_ald_thin', D(40, 32), masks, [
    dep('all', 3, 20, 0), dep('all', 4, 1, 0), dep('all', 5, 0.5, 0), dep('all', 4, 0.7, 0), dep('all', 5, 1.3, 0), dep('all', 4, 0.4, 0),
    dep('all', 5, 9, 0), etc('trench', 255, 20, 0, true), wet(4, 4), ald(6, 1.1, { dt: 0.5 }), ald(7, 2.9), wet(5, 1.2, { dt: 0.4 }),
    ald(3, 0.3), wet(255, 0.7),
  ]);
}

// 7b. single-column accessors (cg1:174-218) mixed with steps: layers resting on the top, layers that
//     leave a gap (the next conformal deposit turns it into a cavity), a removed layer whose slot
//     is reused before the dump, growth through addLayer on a full column (Lmax 2 -> 18)
{
  const add = (i, j, mat, z0, z1) => ({ op: 'addLayer', i, j, mat, z0, z1 });
  const rem = (i, j) => ({ op: 'removeTopLayer', i, j });
  runCase('accessors_small', Object.assign(D(48, 40), { Lmax: 2 }), { half: { objects: [rect(0, 0, 20, 100)] } }, [
    rem(0, 0), dep('all', 3, 10, 0), add(3, 4, 4, 10, 12.34), add(3, 5, 4, 12, 15), add(3, 5, 5, 15, 15.05), add(3, 5, 6, 15.05, 16),
    add(10, 10, 3, 10, 11), add(23, 19, 5, 14.96, 18.04), rem(3, 4), add(3, 4, 7, 10, 11.5), rem(10, 10), add(10, 10, 4, 10, 10.5),
    dep(null, 4, 4, 1.0, true), etc('half', 255, 3, 0), rem(23, 19), add(23, 19, 6, 30, 31), etc(null, 4, 2, 1.0),
    { op: 'strip_connected', mat: 6 }, { op: 'cmp', t: 3, mat: 0 },
  ]);
}

// 8. kernel-cache trap: thicknesses within 1e-3 share one cached kernel (and its baseAdd)
{
  runCase('kcache_small', D(48, 48), { half: { objects: [rect(0, 0, 20, 100)] } }, [
    dep('all', 3, 10, 0), etc('half', 3, 4, 0), dep(null, 4, 6.0, 1.0, true), dep(null, 5, 6.0004, 1.0, true),
    etc(null, 255, 6.0009, 1.0, true),
  ]);
}

// 8b. cells that are not square.  The reference pads its height map by Rx in both directions (cg1:471, 666), so a
//     lateral step is well defined when the taps reach no further along y than along x: dy > dx (Ry <= Rx), or
//     dy < dx at thicknesses where the two radii coincide.
{
  const masks = {
    trench: { objects: [rect(0, 0, 300, 300), rect(30, 0, 16, 300, neg), rect(70, 21, 24, 36, neg)] },
    island: { objects: [rect(0, 0, 300, 300), circle(40, 30, 18, neg)] },
  };
  runCase('unequal_pitch', D(120, 90, 2, 3), masks, [
    dep('all', 3, 30, 0), dep('all', 4, 20, 0), etc('trench', 4, 20, 0), etc('trench', 3, 12, 0, true),
    dep('all', 5, 10, 1.0, true), etc(null, 5, 6, 1.0, true), dep('island', 6, 7.3, 0.5, true),
    etc('island', 255, 9, 0.7, true), dep(null, 4, 4, 1.0, true), etc('all', 4, 14, 1.0, true), dep(null, 7, 2.9, 1.0),
  ]);
  runCase('unequal_pitch_t', D(120, 90, 3, 2), masks, [
    dep('all', 3, 30, 0), dep('all', 4, 20, 0), etc('trench', 4, 20, 0), etc('trench', 3, 12, 0, true),
    dep('all', 5, 4, 1.0, true), etc(null, 5, 2, 1.0, true), dep('island', 6, 3.5, 0.8, true),
    etc('island', 255, 3.9, 0.6, true), dep(null, 4, 2, 1.0),
  ]);
}

// 9. C1 of the benchmark configs through the reference RUNTIME (rt:183-299), 256^2 and 64^2
for (const NX of [256, 64]) {
  const sc = NX / 256;
  const maskData = { objects: [rect(100 * sc, 100 * sc, 200 * sc, 120 * sc), circle(300 * sc, 300 * sc, 60 * sc),
    rect(140 * sc, 130 * sc, 40 * sc, 40 * sc, Object.assign({ angle: 30 }, neg))] };
  const materialColor = { Si: { color: '#888888', id: 3 }, Ox: { color: '#4444ff', id: 4 }, Nit: { color: '#44ff44', id: 5 } };
  const maskList = [{ id: 'mask0000000001', name: 'M1', data: maskData, thumbnail: '' }];
  const processes = [
    { id: 'p1', kind: 'SUBSTR', mask: '-', material: 'Si', thickness: 30, name: 'substrate' },
    { id: 'p2', kind: 'DEPO', mask: '-', material: 'Ox', thickness: 20, name: 'oxide', conformality: 0 },
    { id: 'p3', kind: 'ETCH', mask: 'mask0000000001', material: 'Ox', thickness: 20, name: 'pattern', conformality: 0 },
  ];
  const rt = H.makeRuntime(ref, { LX: 2 * NX, LY: 2 * NX, dx: 2, dy: 2 }, materialColor, maskList);
  const snapshot = { processes, arrowBoundId: 'p3', materialColor, opts: { typ: 'process', procId: null } };
  ref.listeners['simflow:changed'].slice(-1)[0]({ detail: snapshot });
  const g = rt.grid;
  const fn = H.maskPredicate(ref, maskData);
  const raster = Buffer.alloc(g.NX * g.NY);
  let open = 0, h = 2166136261 >>> 0;
  for (let i = 0; i < g.NX; i++) for (let j = 0; j < g.NY; j++) {
    const v = fn((i + 0.5) * 2, (j + 0.5) * 2) ? 1 : 0; raster[i * g.NY + j] = v; open += v;
    h = Math.imul(h ^ v, 16777619) >>> 0;
  }
  const hist = {};
  for (let c = 0; c < g.NX * g.NY; c++) hist[g.cols.len[c]] = (hist[g.cols.len[c]] || 0) + 1;
  const out = { name: 'runtime_c1_' + NX, nx: g.NX, ny: g.NY, domain: { LX: 2 * NX, LY: 2 * NX, dx: 2, dy: 2 },
    materialColor, maskList, processes, snapshot_opts: snapshot.opts, arrowBoundId: 'p3',
    reduced: reduceMask(maskData), fnv: H.gridHash(g), lmax: g.Lmax, open_cells: open,
    raster_fnv: ('00000000' + h.toString(16)).slice(-8), len_hist: hist,
    cache_ids: Object.keys(g.colsCache) };
  fs.writeFileSync(path.join(OUT, out.name + '.json'), JSON.stringify(out));
  if (NX === 64) fs.writeFileSync(path.join(OUT, out.name + '.bin.gz'), zlib.gzipSync(dumpGrid(g), { level: 9 }));
  console.log(out.name, out.fnv, 'open', open, 'raster', out.raster_fnv, JSON.stringify(hist));
}

// 10. conformal kernel taps (cg1:324-357) for a list of (t, S, dx, dy): du, dv, float32 bits of add
{
  const g = new ref.ColumnGrid(8, 8, 2, 2);
  const list = [[10, 1, 2, 2], [6, 1, 2, 2], [7.3, 0.5, 2, 2], [9, 0.7, 2, 2], [2, 1, 2, 2], [30, 1, 2, 2], [30, 0.5, 2, 2],
    [12.7, 0.85, 2, 2], [4, 1, 1, 1], [3.3, 0.9, 0.5, 0.5], [24, 1, 2, 2]];
  const out = [];
  for (const [t, S, dx, dy] of list) {
    g._confKernelCacheArr = new Map();
    const K = g._getConformalKernelArray(t, S, dx, dy);
    const taps = [];
    K.rows.forEach((row, r) => { for (let k = 0; k < row.length; k += 2) {
      const f = new Float32Array([row[k + 1]]); taps.push([r - K.Rx, row[k] | 0, new Uint32Array(f.buffer)[0]]); } });
    out.push({ t, S, dx, dy, Rx: K.Rx, Ry: K.Ry, baseAdd: K.baseAdd, taps });
  }
  fs.writeFileSync(path.join(OUT, 'kernel_taps.json'), JSON.stringify(out));
  console.log('kernel_taps', out.map(o => o.taps.length).join(','));
}

// 11. quantisation known answers (cg1:117-122): round(z/0.1) and q*0.1 for awkward values
{
  const g = new ref.ColumnGrid(8, 8, 2, 2);
  const zs = [0, 0.04, 0.05, 0.15, 0.25, 0.35, 1.45, 2.5, 29.95, 30.05, 57.25, 100.15, 6553.5, 6553.6, 7000.3, 0.049999999999999996, 12.349999999999999];
  const out = zs.map(z => ({ z, q: g._quantizeZ(z), u16: new Uint16Array([g._quantizeZ(z)])[0] }));
  const deq = [0, 1, 3, 7, 300, 305, 12345, 65535].map(q => ({ q, z: g._dequantizeZ(q), requant: g._quantizeZ(g._dequantizeZ(q)) }));
  fs.writeFileSync(path.join(OUT, 'quantize.json'), JSON.stringify({ quant: out, dequant: deq,
    nextpow2: [0, 1, 2, 3, 16, 31, 32, 33, 48].map(n => [n, g._nextPowerOf2(n)]) }));
}

// 12. the runtime's replay / caching policy (rt:183-299): the sequence of grid calls the
//     reference runtime makes during a scripted editing session, recorded with a fake grid
{
  const { makeFakeGridClass, session } = require('../tests/js/fake_grid.js');
  const log = [];
  const S = session();
  ref.window.prj.gridRenderer = { updateFromGrid() { log.push(['render']); } };
  ref.window.prj.ColumnGrid = makeFakeGridClass(log);
  ref.window.prj.processFlow = { materialColor: S.materialColor };
  ref.window.prj.maskmanager = { maskList: S.maskList };
  const rt = new ref.ProcessRuntime({ LX: 200, LY: 200, dx: 2, dy: 2 });
  const marks = [];
  for (const [snapshot, opts] of S.events) {
    marks.push(log.length);
    let err = null;
    try { rt._build(snapshot, opts); } catch (e) { err = e.constructor.name; }
    if (err) log.push(['throw', err]);
  }
  fs.writeFileSync(path.join(OUT, 'runtime_trace.json'), JSON.stringify({ log, marks }));
  console.log('runtime_trace', log.length, 'calls');
}
