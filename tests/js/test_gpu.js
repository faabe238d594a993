// Node-side GPU tests: the golden cases of the reference through the JavaScript ColumnGrid shim
// (N-API addon -> C ABI -> HIP), and config C1 through the flow driver.
'use strict';
const fs = require('fs'), path = require('path'), zlib = require('zlib'), assert = require('assert');
const ROOT = path.join(__dirname, '..', '..');
const GOLD = path.join(ROOT, 'tests', 'golden');
const { ColumnGrid, maskFunctionFor } = require(path.join(ROOT, 'fem_amd', 'js', 'columngrid.js'));
const { ProcessRuntime, isPointOpen } = require(path.join(ROOT, 'fem_amd', 'js', 'runtime.js'));

function fnv(grid) {
  let h = 2166136261 >>> 0;
  const mix = v => { h = Math.imul(h ^ v, 16777619) >>> 0; };
  const { mat, zpair, len } = grid.cols; const L = grid.Lmax;
  for (let c = 0; c < grid.NX * grid.NY; c++) {
    const l = len[c]; mix(l);
    for (let k = 0; k < l; k++) { const b = c * L + k; mix(mat[b]); mix(zpair[2 * b]); mix(zpair[2 * b + 1]); }
  }
  return ('00000000' + h.toString(16)).slice(-8);
}
const same = (a, b, what) => { assert.strictEqual(a.length, b.length, what + ' length'); for (let k = 0; k < a.length; k++) if (a[k] !== b[k]) assert.fail(`${what} differs at ${k}: ${a[k]} vs ${b[k]}`); };

for (const name of ['vertical_small', 'stencil_small', 'stencil_odd', 'masks_small', 'growth_small', 'cavity_small', 'kcache_small', 'anchor_256',
  'wet_ald_small', 'wet_ald_cache', 'wet_ald_thin', 'accessors_small', 'unequal_pitch', 'unequal_pitch_t']) {
  const c = JSON.parse(fs.readFileSync(path.join(GOLD, name + '.json')));
  const binPath = path.join(GOLD, name + '.bin.gz');
  const bin = fs.existsSync(binPath) ? zlib.gunzipSync(fs.readFileSync(binPath)) : null;
  const d = c.domain;
  const g = new ColumnGrid(d.LX, d.LY, d.dx, d.dy, d.Lmax === undefined ? 32 : d.Lmax, d.zScale === undefined ? 0.1 : d.zScale);
  assert.strictEqual(g.NX, c.nx); assert.strictEqual(g.NY, c.ny);
  const n = g.NX * g.NY;
  let off = 0;
  c.steps.forEach((s, idx) => {
    // odd steps use the device rasteriser (tagged closure), even ones a plain closure evaluated per cell
    let fn = null;
    if (s.mask === 'all') fn = () => true;
    else if (s.mask) fn = (idx % 2) ? maskFunctionFor(c.masks[s.mask]) : ((x, y) => isPointOpen(x, y, c.masks[s.mask]));
    switch (s.op) {
      case 'deposit_general': g.deposit_general(fn, s.mat, s.t, s.S); break;
      case 'etch_general': g.etch_general(fn, s.mat, s.t, s.S); break;
      case 'cmp': g.cmp(s.t, s.mat); break;
      case 'strip_connected': g.strip_connected(s.mat); break;
      case 'identify_cavity': g.identify_cavity(); break;
      case 'etch_wet': g.etch_wet(s.mat, s.t, s.dt === undefined ? { isCache: !!s.isCache } : { isCache: !!s.isCache, dt: s.dt }); break;
      case 'deposit_ALD': g.deposit_ALD(s.mat, s.t, s.dt === undefined ? { isCache: !!s.isCache } : { isCache: !!s.isCache, dt: s.dt }); break;
      case 'slider_reset': g.sliderCache = {}; break;
      case 'addLayer': g.addLayer(s.i, s.j, s.mat, s.z0, s.z1); break;
      case 'removeTopLayer': g.removeTopLayer(s.i, s.j); break;
      default: throw new Error(s.op);
    }
    if (s.slider) {
      const a = g.sliderCache.aldCache, w = g.sliderCache.wetetchCache;
      assert.deepStrictEqual(a ? Array.from(a.gridWithCavity.keys()) : [], s.slider.ald_keys, `${name} step ${idx} ALD anchors`);
      assert.deepStrictEqual(w ? Array.from(w.grid.keys()) : [], s.slider.wet_keys, `${name} step ${idx} wet anchors`);
      assert.strictEqual(a && a.exposedSegments ? a.exposedSegments.count : 0, s.slider.seeds, `${name} step ${idx} seeds`);
      assert.strictEqual(w ? w.R : 0, s.slider.wet_R, `${name} step ${idx} wet R`);
    }
    if (s.probe) assert.deepStrictEqual([g.getLen(s.i, s.j), g.topMat(s.i, s.j), g.topZ(s.i, s.j)], s.probe, `${name} step ${idx} probe`);
    assert.strictEqual(g.maxHeight(), s.max_height, `${name} step ${idx} maxHeight`);
    if (s.dump_index !== undefined) {
      const lm = c.dumps[s.dump_index].lmax;
      const cols = g.cols;
      assert.strictEqual(g.Lmax, lm, `${name} step ${idx} Lmax`);
      same(cols.len, bin.subarray(off, off + n), `${name} step ${idx} len`); off += n;
      same(cols.mat, bin.subarray(off, off + n * lm), `${name} step ${idx} mat`); off += n * lm;
      same(new Uint8Array(cols.zpair.buffer, cols.zpair.byteOffset, 4 * n * lm), bin.subarray(off, off + 4 * n * lm), `${name} step ${idx} zpair`); off += 4 * n * lm;
    }
    if (n <= 16384 || idx === c.steps.length - 1) assert.strictEqual(fnv(g), s.fnv, `${name} step ${idx} ${s.op}`);
  });
  g.dispose();
}

// config C1 through the flow driver (reference runtime output: tests/golden/runtime_c1_*.json)
for (const nx of [64, 256]) {
  const c = JSON.parse(fs.readFileSync(path.join(GOLD, `runtime_c1_${nx}.json`)));
  const rt = new ProcessRuntime({ domain: c.domain, processFlow: { materialColor: c.materialColor }, maskmanager: { maskList: c.maskList } });
  rt.onSimflowChanged({ processes: c.processes, arrowBoundId: c.arrowBoundId, materialColor: c.materialColor, opts: c.snapshot_opts });
  assert.strictEqual(fnv(rt.grid), c.fnv, 'runtime C1 ' + nx);
  assert.deepStrictEqual(Object.keys(rt.grid.colsCache), c.cache_ids);
  // cache round trip: reload the saved step and replay
  rt.onSimflowChanged({ processes: c.processes, arrowBoundId: c.arrowBoundId, materialColor: c.materialColor, opts: { typ: 'explorer', procId: null } });
  rt.oldUpto = null;
  rt.onSimflowChanged({ processes: c.processes, arrowBoundId: c.arrowBoundId, materialColor: c.materialColor, opts: { typ: 'explorer', procId: null } });
  assert.strictEqual(fnv(rt.grid), c.fnv, 'runtime C1 after cache reload ' + nx);
  // slices: the YZ line at x = 0 equals the same rows of cols
  const i = rt.grid.indexFromX(0), s = rt.grid.slices('YZ', [i]), cols = rt.grid.cols, L = rt.grid.Lmax, NY = rt.grid.NY;
  same(s.len, cols.len.subarray(i * NY, (i + 1) * NY), 'slice len');
  same(s.mat, cols.mat.subarray(i * NY * L, (i + 1) * NY * L), 'slice mat');
  rt.grid.dispose();
}

// ---- one grid over several slabs, driven by the library (cg_group_*): configuration C4's 20 steps through the JavaScript
// class with 2 and 3 slabs on one device must leave the same grid as the undivided one after every step
{
  const NXc = 122, NYc = 72, L = 2 * NXc;
  const obj = (type, left, top, w, h, polarity, extra) => Object.assign({ type, left, top, width: w, height: h, angle: 0, scaleX: 1, scaleY: 1,
    visible: true, data: { polarity } }, extra || {});
  const periodic = (pitch, size, shift, circle) => {
    const objects = [];
    for (let x = shift; x < 2 * NXc; x += pitch) for (let y = shift; y < 2 * NYc; y += pitch)
      objects.push(circle ? obj('circle', x, y, 2 * size, 2 * size, 'positive', { radius: size }) : obj('rect', x, y, size, size, 'positive'));
    return { objects };
  };
  const masks = {
    M: periodic(64, 32, 0, false), Mshift: periodic(64, 32, 16, false), Mc: periodic(48, 12, 0, true),
    Mr: { objects: [obj('rect', 0, 0, 4 * L, 4 * L, 'positive'), obj('rect', 0.30 * L, 0.05 * L, 0.55 * L, 0.35 * L, 'negative', { angle: 30 })] },
  };
  const SI = 3, OX = 4, NIT = 5, M6 = 6, M7 = 7, ALL = 255;
  const D = (mask, mat, t, S) => ({ op: 'd', mask, mat, t, S: S || 0 }), E = (mask, mat, t, S) => ({ op: 'e', mask, mat, t, S: S || 0 });
  const flow = [D(null, SI, 30), D(null, OX, 20), D(null, NIT, 10), E('M', NIT, 10), E('M', OX, 12), D(null, NIT, 10, 1.0), E(null, NIT, 6, 1.0),
    D(null, M6, 8, 0.5), E('Mc', M6, 8), D(null, M7, 15), E('Mshift', ALL, 10), E('Mshift', SI, 5), D(null, OX, 12, 1.0), E('Mr', OX, 4, 1.0),
    D(null, NIT, 6, 0.5), E('Mc', NIT, 6), D(null, M6, 30), E('M', M6, 30, 0.5), D(null, M7, 4, 1.0), E('Mr', ALL, 8),
    { op: 'cmp', t: 6, mat: 0 }, { op: 'strip', mat: OX }, D(null, OX, 5, 1.0), { op: 'wet', mat: M7, t: 3 }, { op: 'ald', mat: NIT, t: 4 }];
  const apply = (g, fns, s) => {
    const fn = s.mask ? fns[s.mask] : null;
    if (s.op === 'd') g.deposit_general(fn, s.mat, s.t, s.S);
    else if (s.op === 'e') g.etch_general(fn, s.mat, s.t, s.S);
    else if (s.op === 'cmp') g.cmp(s.t, s.mat);
    else if (s.op === 'strip') g.strip_connected(s.mat);
    else if (s.op === 'wet') g.etch_wet(s.mat, s.t);
    else g.deposit_ALD(s.mat, s.t);
  };
  const whole = new ColumnGrid(2 * NXc, 2 * NYc, 2, 2);
  const wfns = {}; for (const k of Object.keys(masks)) wfns[k] = maskFunctionFor(masks[k]);
  const want = flow.map(s => { apply(whole, wfns, s); return [whole.checksum(), whole.maxHeight(), whole.Lmax]; });
  whole.saveCache('final');
  for (const devices of [[0, 0], [0, 0, 0]]) {
    const g = new ColumnGrid(2 * NXc, 2 * NYc, 2, 2, 32, 0.1, { devices });
    assert.strictEqual(g.NX, NXc); assert.strictEqual(g.NY, NYc);
    const fns = {}; for (const k of Object.keys(masks)) fns[k] = maskFunctionFor(masks[k]);
    flow.forEach((s, idx) => { apply(g, fns, s); assert.deepStrictEqual([g.checksum(), g.maxHeight(), g.Lmax], want[idx], `${devices.length} slabs, step ${idx} ${JSON.stringify(s)}`); });
    const a = g.cols, b = whole.cols;
    same(a.len, b.len, 'group len'); same(a.mat, b.mat, 'group mat'); same(a.zpair, b.zpair, 'group zpair');
    const tr = g.transport();
    assert.ok(tr.exchanges > 0 && tr.bytes > 0 && tr.rccl === false, 'rows crossed the slab edges by device copies');
    // a plain closure (evaluated per cell on the host, uploaded as one raster of the whole grid) across the slabs
    g.etch_general((x, y) => x + y > 150, ALL, 3, 1.0); whole.etch_general((x, y) => x + y > 150, ALL, 3, 1.0);
    assert.strictEqual(g.checksum(), whole.checksum(), 'host-evaluated mask over slabs');
    whole.loadCache('final');
    g.dispose();
  }
  whole.dispose();
}

// ---- the flow driver walks a mask's objects once, however many steps go through the mask
{
  const { reduceMask } = require(path.join(ROOT, 'fem_amd', 'js', 'maskshapes.js'));
  const objects = [];
  for (let a = 0; a < 128; a++) for (let b = 0; b < 128; b++)
    objects.push({ type: 'rect', left: 4 * a, top: 4 * b, width: 2, height: 2, angle: 0, scaleX: 1, scaleY: 1, visible: true, data: { polarity: 'positive' } });
  assert.strictEqual(objects.length, 16384);
  const maskList = [{ id: 'mask0000000001', name: 'dots', data: { objects } }];
  const materialColor = { Si: { color: '#888', id: 3 }, Ox: { color: '#00f', id: 4 } };
  const processes = [{ id: 'p0', kind: 'SUBSTR', mask: '-', material: 'Si', thickness: 20, name: 's' }];
  for (let k = 1; k < 20; k++)
    processes.push({ id: 'p' + k, kind: k % 2 ? 'ETCH' : 'DEPO', mask: 'mask0000000001', material: k % 2 ? 'ALL' : 'Ox', thickness: 2 + k % 3, conformality: 0, name: 'x' + k });
  const rt = new ProcessRuntime({ domain: { LX: 512, LY: 512, dx: 2, dy: 2 }, processFlow: { materialColor }, maskmanager: { maskList } });
  const before = reduceMask.calls;
  rt.onSimflowChanged({ processes, arrowBoundId: 'p19', materialColor, opts: { typ: 'process', procId: 'p19' } });
  assert.strictEqual(reduceMask.calls - before, 1, '20 steps through one 16 384-shape mask reduce it once');
  assert.strictEqual(rt.grid._maskIds.size, 1, 'and rasterise it once');
  // the editor saves the mask: a new data object, hence a new closure and a new raster
  maskList[0].data = { objects: objects.slice(0, 100) };
  rt.oldUpto = null;
  rt.onSimflowChanged({ processes, arrowBoundId: 'p19', materialColor, opts: { typ: 'maskchange', procId: 'p1' } });
  assert.strictEqual(reduceMask.calls - before, 2, 'a saved mask is reduced again, once');
  rt.grid.dispose();
}
console.log('js gpu tests ok');
