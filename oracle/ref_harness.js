// TEST INFRASTRUCTURE ONLY -- loads the reference JavaScript, unmodified and in
// place, into a Node `vm` context so that its ColumnGrid and ProcessRuntime can
// be driven headlessly.  Nothing of the reference is copied: the files are read
// as text from REF_DIR at run time (this container only; the reference does not
// travel to the GPU box, so only tests/golden/ fixtures produced with this
// harness are used there).
//
// Run with:  node --harmony-nullish --harmony-optional-chaining <script using this>
//
// The column-grid class is shipped as three files cut at 1000-line boundaries
// (script_columngrid.js, script_columngrid2.js, script_columngrid3.js); the
// unit that parses is their concatenation.
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');

const REF_DIR = process.env.FEM_REFERENCE_DIR || '/root/reference/template';

function loadReference() {
  const listeners = {};
  const windowObj = {
    prj: {},
    addEventListener(name, fn) { (listeners[name] = listeners[name] || []).push(fn); },
    dispatchEvent(ev) { (listeners[ev.type] || []).forEach(fn => fn(ev)); },
  };
  const sandbox = { window: windowObj, console, Math, Number, Object, Array, Map, Set,
    Uint8Array, Uint16Array, Uint32Array, Float32Array, Float64Array, ArrayBuffer };
  sandbox.globalThis = sandbox;
  const ctx = vm.createContext(sandbox);
  const read = f => fs.readFileSync(path.join(REF_DIR, f), 'utf8');
  const cg = read('script_columngrid.js') + '\n' + read('script_columngrid2.js') + '\n' +
             read('script_columngrid3.js') + '\n;globalThis.ColumnGrid = ColumnGrid;';
  vm.runInContext(cg, ctx, { filename: 'columngrid(1+2+3).js' });
  const rt = read('script_runtime.js') + '\n;globalThis.ProcessRuntime = ProcessRuntime;';
  vm.runInContext(rt, ctx, { filename: 'script_runtime.js' });
  return { ColumnGrid: sandbox.ColumnGrid, ProcessRuntime: sandbox.ProcessRuntime,
           window: windowObj, listeners };
}

// Build a headless ProcessRuntime: stubs for the renderer / flow / mask manager globals
// the runtime constructor and _applyStep read (script_runtime.js:9, 17, 48, 153).
function makeRuntime(ref, domain, materialColor, maskList) {
  ref.window.prj.gridRenderer = { updateFromGrid() {} };
  ref.window.prj.ColumnGrid = ref.ColumnGrid;
  ref.window.prj.processFlow = { materialColor };
  ref.window.prj.maskmanager = { maskList };
  return new ref.ProcessRuntime(domain);
}

// Reference mask predicate without constructing a runtime (script_runtime.js:54-71).
function maskPredicate(ref, maskData) {
  const self = Object.create(ref.ProcessRuntime.prototype);
  return (x, y) => ref.ProcessRuntime.prototype._isPointBlocked.call(self, x, y, maskData);
}

// FNV-1a/32 over, per column in order: len, then (mat, z0q, z1q) of each live layer.
function gridHash(grid) {
  let h = 2166136261 >>> 0;
  const mix = v => { h = Math.imul(h ^ v, 16777619) >>> 0; };
  const { mat, zpair, len } = grid.cols;
  const L = grid.Lmax, N = grid.NX * grid.NY;
  for (let c = 0; c < N; c++) {
    const l = len[c]; mix(l);
    for (let k = 0; k < l; k++) { const b = c * L + k; mix(mat[b]); mix(zpair[2 * b]); mix(zpair[2 * b + 1]); }
  }
  return ('00000000' + h.toString(16)).slice(-8);
}

module.exports = { loadReference, makeRuntime, maskPredicate, gridHash, REF_DIR };
