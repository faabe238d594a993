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
console.log('js gpu tests ok');
