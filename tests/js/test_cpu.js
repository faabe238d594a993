// Node-side tests that need no GPU: the addon loads and exports the whole surface, the host mask
// predicate and the shape reduction reproduce the reference's rasters, and the flow driver makes
// the same grid calls as the reference runtime did in the recorded session.
'use strict';
const fs = require('fs'), path = require('path'), zlib = require('zlib'), assert = require('assert');
const ROOT = path.join(__dirname, '..', '..');
const GOLD = path.join(ROOT, 'tests', 'golden');
const addon = require(path.join(ROOT, 'fem_amd', 'js', 'colgrid_napi.node'));
const { ProcessRuntime, isPointOpen } = require(path.join(ROOT, 'fem_amd', 'js', 'runtime.js'));
const { reduceMask } = require(path.join(ROOT, 'fem_amd', 'js', 'maskshapes.js'));
const { makeFakeGridClass, session } = require('./fake_grid.js');

// 1. addon surface
const want = ['create', 'createSlab', 'destroy', 'reset', 'clear', 'info', 'sync', 'maskFromRaster', 'maskFromShapes', 'maskDownload',
  'maskDestroy', 'depositGeneral', 'etchGeneral', 'cmp', 'stripConnected', 'identifyCavity', 'maxHeight', 'download', 'upload',
  'snapshotSave', 'snapshotLoad', 'snapshotDrop', 'slices', 'indexFromPos', 'checksum', 'version'];
for (const f of want) assert.strictEqual(typeof addon[f], 'function', 'addon.' + f);
assert.ok(addon.version().startsWith('colgrid'));
assert.strictEqual(addon.shapeBytes(), require(path.join(ROOT, 'fem_amd', 'js', 'maskshapes.js')).SHAPE_BYTES, 'cg_shape size');

// 2. host predicate + reduction against the reference's rasters
for (const name of ['masks_small', 'stencil_odd', 'vertical_small', 'stencil_small', 'cavity_small']) {
  const c = JSON.parse(fs.readFileSync(path.join(GOLD, name + '.json')));
  const rasters = zlib.gunzipSync(fs.readFileSync(path.join(GOLD, name + '.masks.gz')));
  const n = c.nx * c.ny;
  c.mask_order.forEach((m, k) => {
    const data = c.masks[m];
    for (let i = 0; i < c.nx; i++) for (let j = 0; j < c.ny; j++) {
      const got = isPointOpen((i + 0.5) * c.domain.dx, (j + 0.5) * c.domain.dy, data) ? 1 : 0;
      assert.strictEqual(got, rasters[k * n + i * c.ny + j], `${name}/${m} cell ${i},${j}`);
    }
    assert.deepStrictEqual(reduceMask(data), c.reduced[m], `${name}/${m} reduction`);
  });
}

// 3. replay / caching policy: same call sequence as the reference runtime
{
  const gold = JSON.parse(fs.readFileSync(path.join(GOLD, 'runtime_trace.json')));
  const log = [];
  const S = session();
  const rt = new ProcessRuntime({ domain: { LX: 200, LY: 200, dx: 2, dy: 2 }, ColumnGrid: makeFakeGridClass(log),
    gridRenderer: { updateFromGrid() { log.push(['render']); } },
    processFlow: { materialColor: S.materialColor }, maskmanager: { maskList: S.maskList } });
  const marks = [];
  for (const [snapshot, opts] of S.events) {
    marks.push(log.length);
    let err = null;
    try { rt._build(snapshot, opts); } catch (e) { err = e.constructor.name; }
    if (err) log.push(['throw', err]);
  }
  assert.deepStrictEqual(marks, gold.marks, 'event boundaries');
  assert.strictEqual(log.length, gold.log.length);
  for (let k = 0; k < log.length; k++) assert.deepStrictEqual(JSON.parse(JSON.stringify(log[k])), gold.log[k], 'call ' + k);
}
console.log('js cpu tests ok');
