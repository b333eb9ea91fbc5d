// ESM loader hook for Node 12.x: lets the golden-vector generator import the
// reference codec straight from /root/reference (read-only, never copied).
// The reference targets Node >= 20 and uses `?.` / `??` in 15 guard
// expressions of the form `obj?.field ?? throwError(...)`; Node 12's parser
// rejects them, so the source text is rewritten IN MEMORY at import time:
//   a?.b  -> a.b      a?.[x] -> a[x]      x ?? y -> x || y
// On the happy path (objects present) the semantics are unchanged.
// Nothing is written to disk. This file is test infrastructure only.
export async function transformSource(source, context, defaultTransformSource) {
  const { url } = context
  if (url.includes('/root/reference/') && url.endsWith('.js')) {
    let text = typeof source === 'string' ? source : Buffer.from(source).toString('utf8')
    text = text.replace(/\?\.\[/g, '[').replace(/\?\.(?=[A-Za-z_$])/g, '.').replace(/\s\?\?\s/g, ' || ')
    return { source: text }
  }
  return defaultTransformSource(source, context, defaultTransformSource)
}
